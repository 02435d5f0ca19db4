"""Second, independent restatement of the reference's canonical-Huffman stage -- pure Python, class by class.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): nothing in the product imports it.  It exists because the
reference ships no fixture with canonical-Huffman bytes, so the C restatement (gvrs_oracle_canon.c) could only be checked
against itself.  This module was written from the Java sources alone -- one Python class per Java class, same fields, same
loops, same order of operations, Java `int` / `long` arithmetic made explicit -- WITHOUT consulting the C restatement; the
golden vectors it generates (tests/golden/make_canon_vectors.py -> tests/golden/canon_vectors.json) must be reproduced
byte for byte by the C oracle (tests/test_oracle_canon_vectors.py) and by the GPU (tests/test_gpu_canon_vectors.py).
Two restatements by different routes agreeing is the strongest check available here; it is still not a reference
fixture: rows a13 / f1 of SURVEY.md section 8 stay "unpinned by reference fixtures" (DESIGN.md section 2).

Reference (core/src/main/java/org/gridfour/):
  io/BitOutputStore.java:205-288, io/BitInputStore.java:112-210
  compress/canonicalHuffman/SymbolNode.java:46-113, HuffmanCodeBits.java:46-73, TreeBuilder.java:75-321,
  PackageMerge.java:91-175, LengthEncoder.java:86-236, CanonHuffTreeDecoder.java:68-200,
  CanonicalHuffman.java:177-519, CodecCanonHuffman.java:78-196
  compress/PredictorModel{Differencing,Linear,Triangle,DifferencingWithNulls}.java encodeInt / decodeInt

Pure-Python loops: meant for the small cases of the vector generator (a 120x150 tile takes a few seconds).
"""
import functools
import math

INT4_NULL_CODE = -(2 ** 31)          # util/GridfourConstants.java:61
INT_MAX = 2 ** 31 - 1


def i32(x):
    """Java int: wrap to 32 bits, signed."""
    x &= 0xFFFFFFFF
    return x - (1 << 32) if x & 0x80000000 else x


def i8(x):
    """Java (byte) cast."""
    x &= 0xFF
    return x - 256 if x & 0x80 else x


# ------------------------------------------------------------------ io/BitOutputStore.java
class BitOutputStore:
    """Bits are appended LSB first into 64-bit words that are stored little-endian (BitOutputStore.java:205-288)."""

    def __init__(self):
        self.marker = 1
        self.scratch = 0
        self.nBits = 0
        self.bytes = bytearray()             # the ByteBuffer of :72-151 (block list flattened)

    def _move_scratch_to_text(self):         # :300-304 + ByteBuffer.addLong :91-107
        s = self.scratch
        for _ in range(8):
            self.bytes.append(s & 0xFF)
            s >>= 8
        self.scratch = 0
        self.marker = 1

    def appendBit(self, value):              # :205-215
        if value != 0:
            self.scratch |= self.marker
        self.marker = (self.marker << 1) & 0xFFFFFFFFFFFFFFFF
        self.nBits += 1
        if self.marker == 0:
            self._move_scratch_to_text()

    def appendBits(self, nBitsInValue, value):   # :225-262
        if nBitsInValue < 1 or nBitsInValue > 32:
            raise ValueError("Attempt to add number of bits not in range (1, 32): %d" % nBitsInValue)
        v = value & ((1 << nBitsInValue) - 1)
        nBitsInScratch = self.nBits & 0x3F
        nFreeInScratch = 64 - nBitsInScratch
        if nFreeInScratch < nBitsInValue:
            nBitsShort = nBitsInValue - nFreeInScratch
            lowPart = v & ((1 << nFreeInScratch) - 1)
            highPart = v >> nFreeInScratch
            self.scratch |= lowPart << nBitsInScratch
            self.nBits += nFreeInScratch
            self._move_scratch_to_text()
            self.scratch = highPart
            self.nBits += nBitsShort
            self.marker = 1 << nBitsShort
        else:
            self.scratch |= v << nBitsInScratch
            self.nBits += nBitsInValue
            self.marker = (self.marker << nBitsInValue) & 0xFFFFFFFFFFFFFFFF
            if self.marker == 0:
                self._move_scratch_to_text()

    def getEncodedText(self):                # :269-286
        nBytesToEncode = (self.nBits + 7) // 8
        b = bytearray(self.bytes)
        if len(b) < nBytesToEncode:
            b.extend(bytes(nBytesToEncode - len(b)))
        nBitsInScratch = self.nBits & 0x3F
        if nBitsInScratch > 0:
            s = self.scratch
            iByte = len(self.bytes)
            for _ in range((nBitsInScratch + 7) // 8):
                b[iByte] = s & 0xFF
                iByte += 1
                s >>= 8
        return bytes(b)

    def getEncodedTextLength(self):          # :288-290
        return self.nBits


# ------------------------------------------------------------------ io/BitInputStore.java
class BitInputStore:
    """:64-218.  Reading past the end raises IndexError (Java: ArrayIndexOutOfBoundsException)."""

    def __init__(self, data, offset=0, length=None):
        if length is None:
            length = len(data)
        if length + offset > len(data):
            raise ValueError("Insufficient input.length")
        self.text = bytes(data)
        self.nBits = length * 8
        self.nBytesProcessed = offset
        self.scratch = 0
        self.iBit = 0
        self.nBitsInScratch = 0

    def _move_text_to_scratch(self):         # :180-209
        t, p = self.text, self.nBytesProcessed
        if p + 8 <= len(t):
            self.scratch = int.from_bytes(t[p:p + 8], "little")
            self.nBytesProcessed += 8
            self.nBitsInScratch = 64
        else:
            k = 0
            self.scratch = 0
            for i in range(len(t) - 1, p - 1, -1):
                self.scratch = (self.scratch << 8) | t[i]
                k += 1
            self.nBytesProcessed += k
            self.nBitsInScratch = k * 8

    def getBit(self):                        # :112-125
        if self.nBitsInScratch == 0:
            if self.iBit >= self.nBits:
                raise IndexError("Attempt to read past end of data")
            self._move_text_to_scratch()
        bit = self.scratch & 1
        self.scratch >>= 1
        self.nBitsInScratch -= 1
        self.iBit += 1
        return bit

    def getBits(self, nBitsInValue):         # :136-177
        if nBitsInValue < 1 or nBitsInValue > 32:
            raise ValueError("Attempt to get a number of bits not in range [1..32]")
        if self.nBitsInScratch >= nBitsInValue:
            v = self.scratch & ((1 << nBitsInValue) - 1)
            self.scratch >>= nBitsInValue
            self.nBitsInScratch -= nBitsInValue
            self.iBit += nBitsInValue
            return v
        if self.iBit + nBitsInValue > self.nBits:
            raise IndexError("Attempt to read past end of data")
        v = self.scratch
        nBitsShort = nBitsInValue - self.nBitsInScratch
        nBitsCopied = self.nBitsInScratch
        self._move_text_to_scratch()
        v |= (self.scratch & ((1 << nBitsShort) - 1)) << nBitsCopied
        self.scratch >>= nBitsShort
        self.nBitsInScratch -= nBitsShort
        self.iBit += nBitsInValue
        return i32(v)

    def getPosition(self):
        return self.iBit


# ------------------------------------------------------------------ canonicalHuffman/SymbolNode.java
class SymbolNode:
    """:46-113"""

    def __init__(self, symbol=None, left=None, right=None):
        self.count = 0
        self.bit = 0
        self.nBitsInCode = 0
        self.code = None
        self.next = None
        self.left = None
        self.right = None
        if left is not None:                 # SymbolNode(left, right) :82-90
            self.isLeaf = False
            self.symbol = -1
            self.left = left
            self.right = right
            self.count = right.count + left.count
            left.bit = 0
            right.bit = 1
        elif symbol is None:                 # SymbolNode() :72-75
            self.isLeaf = False
            self.symbol = -1
        else:                                # SymbolNode(int) :77-80
            self.isLeaf = True
            self.symbol = symbol

    def clear(self):                         # :62-70
        self.count = 0
        self.bit = 0
        self.next = None
        self.left = None
        self.right = None
        self.nBitsInCode = 0
        self.code = None


# ------------------------------------------------------------------ canonicalHuffman/HuffmanCodeBits.java
class HuffmanCodeBits:
    """:46-73 -- the canonical successor rule: next code = (previous + 1) << (length growth)."""

    def __init__(self, length, source=None):
        if source is None:                   # HuffmanCodeBits(int) :51-54
            self.nBitsInCode = length
            self.bits = 0
        else:                                # HuffmanCodeBits(source, length) :56-64
            self.bits = source.bits + 1
            if length > source.nBitsInCode:
                self.bits = (self.bits << (length - source.nBitsInCode)) & 0xFFFFFFFFFFFFFFFF
                self.nBitsInCode = length
            else:
                self.nBitsInCode = source.nBitsInCode

    def getCodeBytes(self):                  # :66-72: most significant code bit first into an LSB-first store
        bitpath = BitOutputStore()
        for i in range(self.nBitsInCode - 1, -1, -1):
            bitpath.appendBit((self.bits >> i) & 1)
        return bitpath.getEncodedText()


# ------------------------------------------------------------------ canonicalHuffman/LengthEncoder.java
MAX_STANDARD_SYMBOL = 15
REPEAT_PREV_2BITS = 16
REPEAT_ZERO_3BITS = 17
REPEAT_ZERO_7BITS = 18
SYMBOL_SET_SIZE = 19


class LengthEncoder:
    """:44-236"""

    def __init__(self, nCodedSymbols, nCodes, codes, runLengths):
        self.nCodedSymbols = nCodedSymbols
        self.nCodes = nCodes
        self.codes = list(codes[:nCodes])
        self.runLengths = list(runLengths[:nCodes])

    @staticmethod
    def encodeLengths(n, codeLen):           # :86-166
        countCode = [0] * n
        runLength = [0] * n
        prior = -1
        nCountCode = 0
        iCodeLen = 0
        while iCodeLen < n:                  # a Java for-loop whose body also moves iCodeLen
            if codeLen[iCodeLen] > MAX_STANDARD_SYMBOL:
                raise ValueError("Invalid code length: %d" % codeLen[iCodeLen])
            if codeLen[iCodeLen] == 0:
                prior = 0
                i = iCodeLen + 1
                while i < n:
                    if codeLen[i] != 0:
                        break
                    i += 1
                nZero = i - iCodeLen
                if nZero == 1:
                    countCode[nCountCode] = 0
                    nCountCode += 1
                elif nZero == 2:
                    countCode[nCountCode] = 0
                    nCountCode += 1
                    countCode[nCountCode] = 0
                    nCountCode += 1
                    iCodeLen += 1
                elif nZero <= 10:
                    countCode[nCountCode] = REPEAT_ZERO_3BITS
                    runLength[nCountCode] = nZero - 3
                    nCountCode += 1
                    iCodeLen = i - 1
                else:
                    if nZero > 138:
                        nZero = 138
                    countCode[nCountCode] = REPEAT_ZERO_7BITS
                    runLength[nCountCode] = nZero - 11
                    nCountCode += 1
                    iCodeLen += nZero - 1
            else:
                if codeLen[iCodeLen] == prior:
                    i = iCodeLen + 1
                    while i < n:
                        if codeLen[i] != prior:
                            break
                        i += 1
                    nPrior = i - iCodeLen
                    if nPrior == 1:
                        countCode[nCountCode] = prior
                        nCountCode += 1
                    elif nPrior == 2:
                        countCode[nCountCode] = prior
                        nCountCode += 1
                        countCode[nCountCode] = prior
                        nCountCode += 1
                        iCodeLen = i - 1
                    else:
                        if nPrior > 6:
                            nPrior = 6
                        countCode[nCountCode] = REPEAT_PREV_2BITS
                        runLength[nCountCode] = nPrior - 3
                        nCountCode += 1
                        iCodeLen += nPrior - 1
                else:
                    prior = codeLen[iCodeLen]
                    countCode[nCountCode] = prior
                    nCountCode += 1
            iCodeLen += 1                    # the for-loop's own increment
        return LengthEncoder(n, nCountCode, countCode, runLength)

    @staticmethod
    def writeEncodedLengths(output, nCodes, codes, runLengths):   # :168-195
        for i in range(nCodes):
            index = codes[i]
            output.appendBits(5, index)
            if index == REPEAT_PREV_2BITS:
                output.appendBits(2, runLengths[i])
            elif index == REPEAT_ZERO_3BITS:
                output.appendBits(3, runLengths[i])
            elif index == REPEAT_ZERO_7BITS:
                output.appendBits(7, runLengths[i])

    @staticmethod
    def readEncodedLengths(input, nSymbols, symbols):             # :197-236 (writes past nSymbols raise, as in Java)
        k = 0
        prior = 0
        while k < nSymbols:
            index = input.getBits(5)
            if index <= MAX_STANDARD_SYMBOL:
                prior = index
                symbols[k] = index
                k += 1
            elif index == REPEAT_PREV_2BITS:
                n = input.getBits(2) + 3
                for _ in range(n):
                    symbols[k] = prior
                    k += 1
            elif index == REPEAT_ZERO_3BITS:
                prior = 0
                n = input.getBits(3) + 3
                for _ in range(n):
                    symbols[k] = 0
                    k += 1
            elif index == REPEAT_ZERO_7BITS:
                prior = 0
                n = input.getBits(7) + 11
                for _ in range(n):
                    symbols[k] = 0
                    k += 1
        return k


# ------------------------------------------------------------------ canonicalHuffman/PackageMerge.java
class _Entry:
    def __init__(self, symbol, count):
        self.symbol = symbol
        self.count = count
        self.nBits = 0


class PackageMerge:
    """:53-175.  `input` is the array the tree builder sorted (count ascending, symbol descending); an entry's `symbol`
    is its INDEX in that array (:94-101)."""

    def merge(self, maxCodeLength, input):
        lst = []
        for iInput, node in enumerate(input):
            if node.count > 0:
                node.nBitsInCode = 0
                node.code = None
                lst.append(_Entry(iInput, node.count))
        lst.sort(key=lambda e: (e.count, e.symbol))                    # :107-113
        base = list(lst)
        entries = [None] * maxCodeLength
        entries[0] = base
        for iDepth in range(1, maxCodeLength):                         # phase 1 :120-147
            ix = entries[iDepth - 1]
            nPair = len(ix) // 2
            pair = [_Entry(-1, ix[iPair * 2].count + ix[iPair * 2 + 1].count) for iPair in range(nPair)]
            k = 0
            iBase = 0
            m = [None] * (len(base) + nPair)
            for iPair in range(nPair):
                while iBase < len(base):
                    if base[iBase].count <= pair[iPair].count:
                        m[k] = base[iBase]
                        k += 1
                        iBase += 1
                    else:
                        break
                m[k] = pair[iPair]
                k += 1
            if base[len(base) - 1].count > pair[nPair - 1].count:
                m[len(m) - 1] = base[len(base) - 1]
            entries[iDepth] = m
        n = len(base) * 2 - 2                                          # phase 2 :151-164
        for iEntry in range(len(entries) - 1, -1, -1):
            nMerged = 0
            ix = entries[iEntry]
            for i in range(n):
                if ix[i].symbol == -1:       # (a None here is Java's NullPointerException)
                    nMerged += 1
                else:
                    ix[i].nBits += 1
            n = nMerged * 2
        for e in base:                                                 # phase 3 :168-172
            input[e.symbol].nBitsInCode = e.nBits


# ------------------------------------------------------------------ canonicalHuffman/TreeBuilder.java
def _count_symbol_comp(o1, o2):              # :100-128: count ascending, symbol DESCENDING
    test = (o1.count > o2.count) - (o1.count < o2.count)
    if test == 0:
        test = (o2.symbol > o1.symbol) - (o2.symbol < o1.symbol)
    return test


class TreeBuilder:
    """:48-321"""

    def __init__(self):
        self.symbolNodes = None
        self.maxCodeLengthLimited = False

    def buildTree(self, symbolNodes):        # :75-188
        self.maxCodeLengthLimited = False
        self.symbolNodes = symbolNodes
        for node in symbolNodes:
            if node.count == 0:
                node.nBitsInCode = 0
        sortNodes = [node for node in symbolNodes if node.count > 0]
        sortNodes.sort(key=functools.cmp_to_key(_count_symbol_comp))
        firstNode = sortNodes[0]
        for i in range(len(sortNodes) - 1):
            sortNodes[i].next = sortNodes[i + 1]
        sortNodes[len(sortNodes) - 1].next = None

        root = None
        while True:                          # :139-169
            left = firstNode
            right = firstNode.next
            firstNode = right.next
            left.next = None
            right.next = None
            branch = SymbolNode(left=left, right=right)
            if firstNode is None:
                root = branch
                break
            elif firstNode.count >= branch.count:
                branch.next = firstNode
                firstNode = branch
            else:
                node = firstNode.next
                prior = firstNode
                while node is not None and node.count < branch.count:
                    prior = node
                    node = node.next
                prior.next = branch
                if node is None:
                    prior.next = branch
                else:
                    branch.next = node

        maxCodeLength = self._establishCodeLengths(root, len(sortNodes))
        if maxCodeLength > MAX_STANDARD_SYMBOL:
            self.maxCodeLengthLimited = True
            PackageMerge().merge(MAX_STANDARD_SYMBOL, sortNodes)

        sortNodes.sort(key=functools.cmp_to_key(_count_symbol_comp))
        for i in range(len(sortNodes) - 1):
            sortNodes[i].next = sortNodes[i + 1]
        sortNodes[len(sortNodes) - 1].next = None
        self.populateCanonicalCodes(sortNodes)
        return maxCodeLength

    def _establishCodeLengths(self, root, nSymbols):   # :200-270: depth of every leaf (explicit stack)
        maxCodeLength = 0
        path = [None] * (nSymbols + 2)
        pathBranch = [0] * (nSymbols + 2)
        path[0] = root
        depth = 1
        while depth > 0:
            index = depth - 1
            pNode = path[index]
            pBranch = pathBranch[index]
            if pBranch == 0:
                if pNode.isLeaf:
                    depth -= 1
                    pNode.nBitsInCode = depth
                    if depth > maxCodeLength:
                        maxCodeLength = depth
                    pathBranch[depth] = 0
                    path[depth] = None
                else:
                    pathBranch[index] = 1
                    pathBranch[depth] = 0
                    path[depth] = pNode.left
                    depth += 1
            elif pBranch == 1:
                pathBranch[index] = 2
                pathBranch[depth] = 0
                path[depth] = pNode.right
                depth += 1
            else:
                pathBranch[index] = 0
                path[index] = None
                depth -= 1
        return maxCodeLength

    def populateCanonicalCodes(self, sortNodes):        # :279-297: order (length, symbol), successor rule
        sortNodes.sort(key=lambda o: (o.nBitsInCode, o.symbol))
        codeBits = [None] * len(sortNodes)
        codeBits[0] = HuffmanCodeBits(sortNodes[0].nBitsInCode)
        for i in range(1, len(sortNodes)):
            codeBits[i] = HuffmanCodeBits(sortNodes[i].nBitsInCode, codeBits[i - 1])
        for i in range(len(sortNodes)):
            sortNodes[i].code = codeBits[i].getCodeBytes()

    def writeOneSymbol(self, output, symbol):           # :299-315
        node = self.symbolNodes[symbol & 0xFFFF]
        _append_code(output, node)
        return True


def _append_code(output, node):
    """The loop shared by CanonicalHuffman.encode :209-221, TreeBuilder.writeOneSymbol and SymbolNode.appendToOutput."""
    nFullBytesInCode = node.nBitsInCode // 8
    for j in range(nFullBytesInCode):
        output.appendBits(8, node.code[j])
    remainder = node.nBitsInCode - nFullBytesInCode * 8
    if remainder > 0:
        test = i8(node.code[nFullBytesInCode])           # a Java byte: the shift below is arithmetic
        for _ in range(nFullBytesInCode * 8, node.nBitsInCode):
            output.appendBit(test & 1)
            test >>= 1


# ------------------------------------------------------------------ canonicalHuffman/CanonHuffTreeDecoder.java
class CanonHuffTreeDecoder:
    """:49-200"""

    def __init__(self, symbolLengths):       # :68-132
        nSymbols = len(symbolLengths)
        symbolNodes = []
        lst = []
        for i in range(nSymbols):
            node = SymbolNode(i)
            node.nBitsInCode = symbolLengths[i]
            symbolNodes.append(node)
            if symbolLengths[i] > 0:
                lst.append(node)
        self.nUniqueSymbols = len(lst)
        sortNodes = sorted(lst, key=lambda o: (o.nBitsInCode, o.symbol))
        codeBits = [None] * len(sortNodes)
        codeBits[0] = HuffmanCodeBits(sortNodes[0].nBitsInCode)        # (an empty list is Java's AIOOBE: IndexError here)
        for i in range(1, len(sortNodes)):
            codeBits[i] = HuffmanCodeBits(sortNodes[i].nBitsInCode, codeBits[i - 1])
        n = nSymbols * 2 + 2
        self.nodeIndex = [-1] * (n * 3)
        nUsed = 3
        minCodeLength = sortNodes[0].nBitsInCode
        self.kLookup = 8 if minCodeLength > 8 else minCodeLength
        self.lookup = [0] * (1 << self.kLookup)
        for iNode in range(len(sortNodes)):
            node = sortNodes[iNode]
            index = 0
            bits = codeBits[iNode].bits
            iLookup = 0
            for k in range(node.nBitsInCode):
                i = node.nBitsInCode - 1 - k
                bit = (bits >> i) & 1
                iLookup |= bit << k
                test = self.nodeIndex[index + 1 + bit]
                if test < 0:
                    self.nodeIndex[index + 1 + bit] = nUsed
                    index = nUsed
                    nUsed += 3
                else:
                    index = test
                if k == self.kLookup - 1:
                    self.lookup[iLookup] = index
            self.nodeIndex[index] = node.symbol

    def decodeTree(self, input, nSymbols, symbols):     # :134-177
        nodeIndex = self.nodeIndex
        prior = 0
        i = 0
        while i < nSymbols:
            offset = nodeIndex[1 + input.getBit()]
            while nodeIndex[offset] == -1:
                offset = nodeIndex[offset + 1 + input.getBit()]
            test = nodeIndex[offset]
            if test <= MAX_STANDARD_SYMBOL:
                symbols[i] = test
                prior = test
            elif test == REPEAT_PREV_2BITS:
                n = input.getBits(2) + 3
                for j in range(n):
                    symbols[i + j] = prior
                i += n - 1
            elif test == REPEAT_ZERO_3BITS:
                prior = 0
                n = input.getBits(3) + 3
                for j in range(n):
                    symbols[i + j] = 0
                i += n - 1
            elif test == REPEAT_ZERO_7BITS:
                prior = 0
                n = input.getBits(7) + 11
                for j in range(n):
                    symbols[i + j] = 0
                i += n - 1
            i += 1
        return True


# ------------------------------------------------------------------ canonicalHuffman/CanonicalHuffman.java
N_SYMBOLS_TOTAL = 260
N_SYMBOLS_STANDARD = 256
I_NULL_DATA_CODE = 256
I_ESCAPE_1BYTE = 257
I_ESCAPE_2BITS = 258
I_END_OF_TEXT = 259


class CanonicalHuffman:
    """:65-519"""

    def __init__(self):
        self.symbolNodes = [SymbolNode(i) for i in range(N_SYMBOLS_TOTAL)]
        self.nUniqueSymbols = 0
        self.nBitsInCodeTable = 0
        self.maxCodeLengthLimited = False

    def clear(self):                         # :119-137
        self.nUniqueSymbols = 0
        self.maxCodeLengthLimited = False
        for node in self.symbolNodes:
            node.clear()

    def encode(self, output, nSymbolsInText, offset, text):      # :177-283
        if self.nUniqueSymbols > 0:
            self.clear()
        if nSymbolsInText <= 0 or offset < 0 or text is None:
            raise ValueError("Empty or null data input data")
        if nSymbolsInText + offset > len(text):
            raise ValueError("Text array too small for offset and symbol-count specifications")
        self.countSymbols(nSymbolsInText, offset, text)
        textTree = TreeBuilder()
        textTree.buildTree(self.symbolNodes)
        self.maxCodeLengthLimited = textTree.maxCodeLengthLimited
        textCodeLengths = [node.nBitsInCode for node in self.symbolNodes]
        self._buildCodeLengthTree(output, textCodeLengths)
        self.nBitsInCodeTable = output.getEncodedTextLength()
        for iSymbol in range(nSymbolsInText):
            symbol = text[iSymbol]           # sic: no offset here (:204); every caller passes offset 0
            if -128 <= symbol <= 127:
                _append_code(output, self.symbolNodes[symbol + 128])
            elif -512 <= symbol <= 511:
                textTree.writeOneSymbol(output, (symbol >> 2) + 128)
                textTree.writeOneSymbol(output, I_ESCAPE_2BITS)
                output.appendBits(2, symbol & 0x03)
            elif -2048 <= symbol <= 2047:
                textTree.writeOneSymbol(output, (symbol >> 4) + 128)
                textTree.writeOneSymbol(output, I_ESCAPE_2BITS)
                output.appendBits(2, (symbol >> 2) & 0x03)
                textTree.writeOneSymbol(output, I_ESCAPE_2BITS)
                output.appendBits(2, symbol & 0x03)
            elif -8192 <= symbol <= 8191:
                textTree.writeOneSymbol(output, (symbol >> 6) + 128)
                textTree.writeOneSymbol(output, I_ESCAPE_2BITS)
                output.appendBits(2, (symbol >> 4) & 0x03)
                textTree.writeOneSymbol(output, I_ESCAPE_2BITS)
                output.appendBits(2, (symbol >> 2) & 0x03)
                textTree.writeOneSymbol(output, I_ESCAPE_2BITS)
                output.appendBits(2, symbol & 0x03)
            elif -32768 <= symbol <= 32767:
                textTree.writeOneSymbol(output, (symbol >> 8) + 128)
                textTree.writeOneSymbol(output, I_ESCAPE_1BYTE)
                output.appendBits(8, symbol & 0xFF)
            elif symbol == INT4_NULL_CODE:
                textTree.writeOneSymbol(output, I_NULL_DATA_CODE)
            elif -8333608 <= symbol <= 8388607:          # sic (:258): countSymbols tests -8388608
                textTree.writeOneSymbol(output, (symbol >> 16) + 128)
                textTree.writeOneSymbol(output, I_ESCAPE_1BYTE)
                output.appendBits(8, (symbol >> 8) & 0xFF)
                textTree.writeOneSymbol(output, I_ESCAPE_1BYTE)
                output.appendBits(8, symbol & 0xFF)
            else:
                textTree.writeOneSymbol(output, (symbol >> 24) + 128)
                textTree.writeOneSymbol(output, I_ESCAPE_1BYTE)
                output.appendBits(8, (symbol >> 16) & 0xFF)
                textTree.writeOneSymbol(output, I_ESCAPE_1BYTE)
                output.appendBits(8, (symbol >> 8) & 0xFF)
                textTree.writeOneSymbol(output, I_ESCAPE_1BYTE)
                output.appendBits(8, symbol & 0xFF)
        textTree.writeOneSymbol(output, I_END_OF_TEXT)
        return output.getEncodedTextLength()

    def _buildCodeLengthTree(self, output, textCodeLengths):     # :285-343
        textCodeLengthPack = LengthEncoder.encodeLengths(len(textCodeLengths), textCodeLengths)
        nodes = [SymbolNode(i) for i in range(SYMBOL_SET_SIZE + 1)]
        nodes[SYMBOL_SET_SIZE].count = 1
        for i in range(textCodeLengthPack.nCodes):
            nodes[textCodeLengthPack.codes[i]].count += 1
        codeTableTree = TreeBuilder()
        codeTableTree.buildTree(nodes)
        codeTableTreeLengths = [node.nBitsInCode for node in nodes]
        codeTableTreeLengthPack = LengthEncoder.encodeLengths(len(codeTableTreeLengths), codeTableTreeLengths)
        output.appendBit(0)                  # one reserved bit
        LengthEncoder.writeEncodedLengths(output, codeTableTreeLengthPack.nCodes, codeTableTreeLengthPack.codes,
                                          codeTableTreeLengthPack.runLengths)
        for i in range(textCodeLengthPack.nCodes):
            code = textCodeLengthPack.codes[i]
            codeTableTree.writeOneSymbol(output, code)
            if code > MAX_STANDARD_SYMBOL:
                runLength = textCodeLengthPack.runLengths[i]
                if code == REPEAT_PREV_2BITS:
                    output.appendBits(2, runLength)
                elif code == REPEAT_ZERO_3BITS:
                    output.appendBits(3, runLength)
                elif code == REPEAT_ZERO_7BITS:
                    output.appendBits(7, runLength)

    def countSymbols(self, nSymbolsInText, offset, text):        # :352-418
        nodes = self.symbolNodes
        nodes[I_END_OF_TEXT].count = 1
        for iSymbol in range(nSymbolsInText):
            symbol = text[iSymbol + offset]
            if -128 <= symbol <= 127:
                nodes[symbol + 128].count += 1
            elif -512 <= symbol <= 511:
                nodes[I_ESCAPE_2BITS].count += 1
                nodes[(symbol >> 2) + 128].count += 1
            elif -2048 <= symbol <= 2047:
                nodes[I_ESCAPE_2BITS].count += 2
                nodes[(symbol >> 4) + 128].count += 1
            elif -8192 <= symbol <= 8191:
                nodes[I_ESCAPE_2BITS].count += 3
                nodes[(symbol >> 6) + 128].count += 1
            elif -32768 <= symbol <= 32767:
                nodes[I_ESCAPE_1BYTE].count += 1
                nodes[(symbol >> 8) + 128].count += 1
            elif symbol == INT4_NULL_CODE:
                nodes[I_NULL_DATA_CODE].count += 1
            elif -8388608 <= symbol <= 8388607:
                nodes[I_ESCAPE_1BYTE].count += 2
                nodes[(symbol >> 16) + 128].count += 1
            else:
                nodes[I_ESCAPE_1BYTE].count += 3
                nodes[(symbol >> 24) + 128].count += 1
        for node in nodes:
            if node.count > 0:
                self.nUniqueSymbols += 1

    def decode(self, input, nSymbolsInText, text):               # :441-466
        if self.nUniqueSymbols > 0:
            self.clear()
        if nSymbolsInText <= 0:
            return False
        input.getBit()
        codeTableLengths = [0] * (SYMBOL_SET_SIZE + 1)
        LengthEncoder.readEncodedLengths(input, SYMBOL_SET_SIZE + 1, codeTableLengths)
        codeTable = CanonHuffTreeDecoder(codeTableLengths)
        textTreeLengths = [0] * (N_SYMBOLS_TOTAL + 1)
        codeTable.decodeTree(input, N_SYMBOLS_TOTAL, textTreeLengths)
        self.nBitsInCodeTable = input.getPosition()
        textTree = CanonHuffTreeDecoder(textTreeLengths)
        self.nUniqueSymbols = textTree.nUniqueSymbols
        self._decodeText(textTree, input, nSymbolsInText, text)
        return True

    def _decodeText(self, textTree, input, nSymbolsInText, text):   # :469-519; ends at the end-of-text symbol
        nodeIndex = textTree.nodeIndex
        lookup = textTree.lookup
        kLookup = textTree.kLookup
        prior = 0
        iSymbol = 0
        while True:
            iX = input.getBits(kLookup)
            offset = lookup[iX]
            while nodeIndex[offset] == -1:
                offset = nodeIndex[offset + 1 + input.getBit()]
            symbol = nodeIndex[offset]
            if symbol == I_END_OF_TEXT:
                break
            if symbol < N_SYMBOLS_STANDARD:
                symbol -= 128
                text[iSymbol] = symbol       # (beyond the array: Java's AIOOBE, IndexError here)
                iSymbol += 1
                prior = symbol
            elif symbol == I_ESCAPE_2BITS:
                part = input.getBits(2)
                prior = i32((prior << 2) | part)
                if iSymbol - 1 < 0:
                    raise IndexError("escape before any value")
                text[iSymbol - 1] = prior
            elif symbol == I_ESCAPE_1BYTE:
                part = input.getBits(8)
                prior = i32((prior << 8) | part)
                if iSymbol - 1 < 0:
                    raise IndexError("escape before any value")
                text[iSymbol - 1] = prior
            elif symbol == I_NULL_DATA_CODE:
                prior = INT4_NULL_CODE
                text[iSymbol] = INT4_NULL_CODE
                iSymbol += 1
        return iSymbol


# ------------------------------------------------------------------ compress/PredictorModel*.java (integer forms)
def _diff_encode(nRows, nColumns, values):           # PredictorModelDifferencing.java:170-200
    out = []
    seed = values[0]
    prior = seed
    for i in range(1, nColumns):
        test = values[i]
        out.append(i32(test - prior))
        prior = test
    for iRow in range(1, nRows):
        index = iRow * nColumns
        prior = values[index - nColumns]
        for _ in range(nColumns):
            test = values[index]
            index += 1
            out.append(i32(test - prior))
            prior = test
    return seed, out


def _linear_encode(nRows, nColumns, values):         # PredictorModelLinear.java:146-185
    out = []
    seed = values[0]
    prior = values[0]
    out.append(i32(values[1] - prior))
    for iRow in range(1, nRows):
        index = iRow * nColumns
        test = values[index]
        out.append(i32(test - prior))
        prior = test
        test = values[index + 1]
        out.append(i32(test - prior))
    for iRow in range(nRows):
        index = iRow * nColumns
        a = values[index]
        b = values[index + 1]
        for iCol in range(2, nColumns):
            c = values[index + iCol]
            prediction = i32(2 * b - a)
            out.append(i32(c - prediction))
            a = b
            b = c
    return seed, out


def _triangle_encode(nRows, nColumns, values):       # PredictorModelTriangle.java:148-186
    if nRows < 2 or nColumns < 2:
        return None, None                    # returns -1: the caller's CanonicalHuffman.encode then throws
    out = []
    seed = values[0]
    prior = seed
    for i in range(1, nColumns):
        test = values[i]
        out.append(i32(test - prior))
        prior = test
    prior = seed
    for i in range(1, nRows):
        test = values[i * nColumns]
        out.append(i32(test - prior))
        prior = test
    for iRow in range(1, nRows):
        k1 = iRow * nColumns
        k0 = k1 - nColumns
        for _ in range(1, nColumns):
            za = values[k0]
            k0 += 1
            zb = values[k1]
            k1 += 1
            zc = values[k0]
            prediction = i32(zc + zb - za)
            out.append(i32(values[k1] - prediction))
    return seed, out


def _nulls_encode(nRows, nColumns, values):          # PredictorModelDifferencingWithNulls.java:169-237
    sumStart = 0
    nStart = 0
    nullFlag = True
    for iRow in range(nRows):
        rowOffset = iRow * nColumns
        for iCol in range(nColumns):
            test = values[rowOffset + iCol]
            if test == INT4_NULL_CODE:
                nullFlag = True
            else:
                if nullFlag:
                    sumStart += test
                    nStart += 1
                nullFlag = False
        nullFlag = values[rowOffset] == INT4_NULL_CODE
    if nStart == 0:
        return 0, []
    avgStart = float(sumStart) / nStart
    f = math.floor(avgStart + 0.5)
    seed = INT_MAX if f >= INT_MAX else (INT4_NULL_CODE if f <= INT4_NULL_CODE else int(f))    # Java (int) of a double saturates
    out = []
    prior = seed
    nullFlag = False
    for iRow in range(nRows):
        index = iRow * nColumns
        for _ in range(nColumns):
            test = values[index]
            index += 1
            if test == INT4_NULL_CODE:
                nullFlag = True
                out.append(INT4_NULL_CODE)
            else:
                if nullFlag:
                    prior = seed
                    nullFlag = False
                out.append(i32(test - prior))
                prior = test
        prior = values[iRow * nColumns]
        nullFlag = prior == INT4_NULL_CODE
    return seed, out


# ------------------------------------------------------------------ canonicalHuffman/CodecCanonHuffman.java
class CodecCanonHuffman:
    """encode :78-148, compress :150-166.  Returns (packing bytes | None, predictor code of the winner)."""

    MODELS = ((1, _diff_encode, False), (2, _linear_encode, False), (3, _triangle_encode, False), (4, _nulls_encode, True))

    def encode(self, codecIndex, nRows, nCols, values):
        values = [int(v) for v in values]
        containsNullValue = any(v == INT4_NULL_CODE for v in values)
        containsValidData = any(v != INT4_NULL_CODE for v in values)
        if not containsValidData:
            return None, 0
        if all(v == values[0] for v in values[1:]):
            store = BitOutputStore()
            store.appendBits(8, codecIndex)
            store.appendBits(8, 0)
            store.appendBits(32, values[0])
            return store.getEncodedText(), 0
        resultLength = INT_MAX
        result, used = None, 0
        for code, fn, nullSupported in self.MODELS:
            if containsNullValue != nullSupported:
                continue
            seed, residuals = fn(nRows, nCols, values)
            if residuals is None:
                raise ValueError("IllegalArgumentException: Empty or null data input data")     # encode(-1, ...) :183
            padded = residuals + [0] * (nRows * nCols - len(residuals))
            store = BitOutputStore()
            store.appendBits(8, codecIndex)
            store.appendBits(8, code)
            store.appendBits(32, seed)
            header = store.getEncodedText()
            body_store = BitOutputStore()
            CanonicalHuffman().encode(body_store, len(residuals), 0, padded)
            testPacking = header[:6] + body_store.getEncodedText()
            if len(testPacking) < resultLength:
                resultLength = len(testPacking)
                result, used = testPacking, code
        return result, used


def canon_encode_streams(streams):
    """CanonicalHuffman.encode of every int list in `streams` into ONE bit store, as LsEncoder12.java:148-151 does with
    one CanonicalHuffman instance; returns (bytes, bit length after each stream)."""
    ch = CanonicalHuffman()
    store = BitOutputStore()
    ends = []
    for text in streams:
        ends.append(ch.encode(store, len(text), 0, [int(v) for v in text]))
    return store.getEncodedText(), ends


def canon_decode_streams(data, counts):
    """CanonicalHuffman.decode of len(counts) streams in sequence from one bit store (LsDecoder12.java:107-119); counts[i] =
    the nSymbolsInText the reference passes (the capacity of its text array).  Returns (lists, bit positions)."""
    ch = CanonicalHuffman()
    store = BitInputStore(data)
    outs, ends = [], []
    for n in counts:
        text = [0] * n
        ok = ch.decode(store, n, text)
        assert ok
        outs.append(text)
        ends.append(store.getPosition())
    return outs, ends
