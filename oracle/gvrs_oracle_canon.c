/*
 * gvrs_oracle_canon.c -- CPU restatement of Gridfour's canonical-Huffman entropy stage and of
 * CodecCanonHuffman.  TEST INFRASTRUCTURE: see gvrs_oracle.h.
 *
 * PARITY UNPINNED: no fixture of the reference holds canonical-Huffman bytes; this
 * file follows the Java sources statement by statement, including their quirks, and is checked by
 * round trips and by hand-derived small cases only (tests/test_oracle_canon.py).
 *
 * Paths are relative to core/src/main/java/org/gridfour/ in the reference repository.
 */
#include "gvrs_oracle.h"
#include "oracle_bits.h"

#include <stdlib.h>
#include <string.h>

#define N_SYMBOLS_TOTAL 260      /* compress/canonicalHuffman/CanonicalHuffman.java:74-80 */
#define N_SYMBOLS_STANDARD 256
#define I_NULL_DATA_CODE 256
#define I_ESCAPE_1BYTE 257
#define I_ESCAPE_2BITS 258
#define I_END_OF_TEXT 259

#define MAX_STANDARD_SYMBOL 15   /* compress/canonicalHuffman/LengthEncoder.java:49-71 */
#define REPEAT_PREV_2BITS 16
#define REPEAT_ZERO_3BITS 17
#define REPEAT_ZERO_7BITS 18
#define SYMBOL_SET_SIZE 19

#define MAX_ALPHABET 261

/* compress/canonicalHuffman/SymbolNode.java */
typedef struct cnode {
    int isLeaf;
    int symbol;
    int count;
    int nBitsInCode;
    uint64_t codeBits;           /* HuffmanCodeBits.bits: canonical code value, MSB emitted first */
    struct cnode *next, *left, *right;
} cnode_t;

/* TreeBuilder.java:124-128: count ascending, symbol DESCENDING */
static int cmp_count_symdesc(const void *pa, const void *pb)
{
    const cnode_t *a = *(cnode_t *const *)pa, *b = *(cnode_t *const *)pb;
    if (a->count != b->count) return a->count < b->count ? -1 : 1;
    if (a->symbol != b->symbol) return a->symbol > b->symbol ? -1 : 1;
    return 0;
}

/* TreeBuilder.java:283-289, CanonHuffTreeDecoder.java:82-88: code length ascending, symbol ascending */
static int cmp_len_sym(const void *pa, const void *pb)
{
    const cnode_t *a = *(cnode_t *const *)pa, *b = *(cnode_t *const *)pb;
    if (a->nBitsInCode != b->nBitsInCode) return a->nBitsInCode - b->nBitsInCode;
    return a->symbol - b->symbol;
}

/* TreeBuilder.establishCodeLengths :193-274 (depth of every leaf) */
static int establish_lengths(cnode_t *node, int depth)
{
    if (node->isLeaf) {
        node->nBitsInCode = depth;
        return depth;
    }
    int a = establish_lengths(node->left, depth + 1);
    int b = establish_lengths(node->right, depth + 1);
    return a > b ? a : b;
}

/* PackageMerge.merge, compress/canonicalHuffman/PackageMerge.java:91-175.  `input` = the sorted node
 * array of TreeBuilder; Entry.symbol = index into it, -1 for a package; a slot the Java code leaves
 * null is marked -2 (dereferencing it would be a NullPointerException there). */
typedef struct { int symbol; int count; int nBits; } pm_entry_t;

static int package_merge(int maxCodeLength, cnode_t **input, int nInput)
{
    pm_entry_t *base = calloc((size_t)nInput, sizeof *base);
    int nBase = 0;
    for (int i = 0; i < nInput; i++) {
        if (input[i]->count > 0) {
            input[i]->nBitsInCode = 0;
            base[nBase].symbol = i;
            base[nBase].count = input[i]->count;
            nBase++;
        }
    }
    /* :110-116 sort by (count asc, index asc): stable insertion sort, the list is nearly sorted */
    for (int i = 1; i < nBase; i++) {
        pm_entry_t e = base[i];
        int j = i - 1;
        while (j >= 0 && (base[j].count > e.count || (base[j].count == e.count && base[j].symbol > e.symbol))) {
            base[j + 1] = base[j];
            j--;
        }
        base[j + 1] = e;
    }
    /* entries[d][i]: index >= 0 -> base entry, -1 -> package, -2 -> null; counts kept beside */
    int **ent = calloc((size_t)maxCodeLength, sizeof *ent);
    int **cnt = calloc((size_t)maxCodeLength, sizeof *cnt);
    int *len = calloc((size_t)maxCodeLength, sizeof *len);
    ent[0] = malloc(sizeof(int) * (size_t)nBase);
    cnt[0] = malloc(sizeof(int) * (size_t)nBase);
    len[0] = nBase;
    for (int i = 0; i < nBase; i++) { ent[0][i] = i; cnt[0][i] = base[i].count; }
    int rc = GVO_OK;
    for (int d = 1; d < maxCodeLength && rc == GVO_OK; d++) {              /* :124-147 */
        const int *ix = ent[d - 1], *ic = cnt[d - 1];
        int nPair = len[d - 1] / 2;
        int mlen = nBase + nPair;
        int *m = malloc(sizeof(int) * (size_t)(mlen > 0 ? mlen : 1));
        int *mc = malloc(sizeof(int) * (size_t)(mlen > 0 ? mlen : 1));
        for (int i = 0; i < mlen; i++) { m[i] = -2; mc[i] = 0; }
        int k = 0, iBase = 0;
        for (int iPair = 0; iPair < nPair; iPair++) {
            if (ix[iPair * 2] == -2 || ix[iPair * 2 + 1] == -2) { rc = GVO_ERR_BOUNDS; break; }  /* NPE in Java */
            int pc = ic[iPair * 2] + ic[iPair * 2 + 1];
            while (iBase < nBase) {
                if (base[iBase].count <= pc) { m[k] = iBase; mc[k] = base[iBase].count; k++; iBase++; }
                else break;
            }
            m[k] = -1; mc[k] = pc; k++;
        }
        if (rc == GVO_OK && nPair > 0) {
            int lastPair = ic[(nPair - 1) * 2] + ic[(nPair - 1) * 2 + 1];
            if (base[nBase - 1].count > lastPair) { m[mlen - 1] = nBase - 1; mc[mlen - 1] = base[nBase - 1].count; }
        } else if (rc == GVO_OK) {
            rc = GVO_ERR_BOUNDS;                                            /* pair[nPair-1] with nPair == 0 */
        }
        ent[d] = m; cnt[d] = mc; len[d] = mlen;
    }
    if (rc == GVO_OK) {                                                     /* :151-165 */
        int n = nBase * 2 - 2;
        for (int d = maxCodeLength - 1; d >= 0 && rc == GVO_OK; d--) {
            int nMerged = 0;
            for (int i = 0; i < n; i++) {
                if (i >= len[d] || ent[d][i] == -2) { rc = GVO_ERR_BOUNDS; break; }
                if (ent[d][i] == -1) nMerged++;
                else base[ent[d][i]].nBits++;
            }
            n = nMerged * 2;
        }
    }
    if (rc == GVO_OK)
        for (int i = 0; i < nBase; i++) input[base[i].symbol]->nBitsInCode = base[i].nBits;   /* :168-172 */
    for (int d = 0; d < maxCodeLength; d++) { free(ent[d]); free(cnt[d]); }
    free(ent); free(cnt); free(len); free(base);
    return rc;
}

/* TreeBuilder.populateCanonicalCodes :283-300 + HuffmanCodeBits.java:47-64 */
static void populate_canonical(cnode_t **sorted, int n)
{
    qsort(sorted, (size_t)n, sizeof *sorted, cmp_len_sym);
    uint64_t bits = 0;
    int curLen = sorted[0]->nBitsInCode;
    sorted[0]->codeBits = 0;
    for (int i = 1; i < n; i++) {
        bits = bits + 1;
        int length = sorted[i]->nBitsInCode;
        if (length > curLen) { bits <<= (length - curLen); curLen = length; }
        sorted[i]->codeBits = bits;
    }
}

/* TreeBuilder.buildTree :75-188: code lengths + canonical codes for nodes[0..nNodes) with count > 0.
 * *limited = package-merge was applied.  Needs >= 2 used symbols (the callers guarantee it). */
static int build_tree(cnode_t *nodes, int nNodes, int *limited)
{
    cnode_t *sorted[MAX_ALPHABET];
    cnode_t branches[MAX_ALPHABET];
    int k = 0, nBranch = 0;
    *limited = 0;
    for (int i = 0; i < nNodes; i++) {
        nodes[i].next = nodes[i].left = nodes[i].right = NULL;
        if (nodes[i].count == 0) nodes[i].nBitsInCode = 0;
        else sorted[k++] = &nodes[i];
    }
    if (k < 2) return GVO_ERR_BOUNDS;                     /* firstNode.next == null -> NPE in Java */
    qsort(sorted, (size_t)k, sizeof *sorted, cmp_count_symdesc);
    for (int i = 0; i < k - 1; i++) sorted[i]->next = sorted[i + 1];
    sorted[k - 1]->next = NULL;
    cnode_t *firstNode = sorted[0], *root = NULL;
    for (;;) {                                            /* :139-169 */
        cnode_t *left = firstNode, *right = firstNode->next;
        firstNode = right->next;
        left->next = right->next = NULL;
        cnode_t *branch = &branches[nBranch++];
        memset(branch, 0, sizeof *branch);
        branch->symbol = -1;
        branch->left = left;
        branch->right = right;
        branch->count = right->count + left->count;
        if (firstNode == NULL) { root = branch; break; }
        if (firstNode->count >= branch->count) {
            branch->next = firstNode;
            firstNode = branch;
        } else {
            cnode_t *node = firstNode->next, *prior = firstNode;
            while (node != NULL && node->count < branch->count) { prior = node; node = node->next; }
            prior->next = branch;
            if (node != NULL) branch->next = node;
        }
    }
    int maxLen = establish_lengths(root, 0);
    if (maxLen > MAX_STANDARD_SYMBOL) {                   /* :173-178 */
        *limited = 1;
        int rc = package_merge(MAX_STANDARD_SYMBOL, sorted, k);
        if (rc != GVO_OK) return rc;
    }
    qsort(sorted, (size_t)k, sizeof *sorted, cmp_count_symdesc);
    populate_canonical(sorted, k);
    return GVO_OK;
}

/* TreeBuilder.writeOneSymbol :302-318: the code bits most-significant first */
static void write_symbol(bitw_t *w, const cnode_t *node)
{
    for (int i = node->nBitsInCode - 1; i >= 0; i--) bw_bit(w, (int)((node->codeBits >> i) & 1u));
}

/* LengthEncoder.encodeLengths, LengthEncoder.java:86-166 */
static int encode_lengths(int n, const int *codeLen, int *codes, int *runLengths)
{
    int prior = -1, i, nCount = 0;
    for (int iCodeLen = 0; iCodeLen < n; iCodeLen++) {
        runLengths[nCount] = 0;
        if (codeLen[iCodeLen] == 0) {
            prior = 0;
            for (i = iCodeLen + 1; i < n; i++) if (codeLen[i] != 0) break;
            int nZero = i - iCodeLen;
            if (nZero == 1) {
                codes[nCount++] = 0;
            } else if (nZero == 2) {
                codes[nCount++] = 0;
                runLengths[nCount] = 0;
                codes[nCount++] = 0;
                iCodeLen++;
            } else if (nZero <= 10) {
                codes[nCount] = REPEAT_ZERO_3BITS;
                runLengths[nCount] = nZero - 3;
                nCount++;
                iCodeLen = i - 1;
            } else {
                if (nZero > 138) nZero = 138;
                codes[nCount] = REPEAT_ZERO_7BITS;
                runLengths[nCount] = nZero - 11;
                nCount++;
                iCodeLen += nZero - 1;
            }
        } else if (codeLen[iCodeLen] == prior) {
            for (i = iCodeLen + 1; i < n; i++) if (codeLen[i] != prior) break;
            int nPrior = i - iCodeLen;
            if (nPrior == 1) {
                codes[nCount++] = prior;
            } else if (nPrior == 2) {
                codes[nCount++] = prior;
                runLengths[nCount] = 0;
                codes[nCount++] = prior;
                iCodeLen = i - 1;
            } else {
                if (nPrior > 6) nPrior = 6;
                codes[nCount] = REPEAT_PREV_2BITS;
                runLengths[nCount] = nPrior - 3;
                nCount++;
                iCodeLen += nPrior - 1;
            }
        } else {
            prior = codeLen[iCodeLen];
            codes[nCount++] = prior;
        }
    }
    return nCount;
}

static int run_bits(int code)
{
    return code == REPEAT_PREV_2BITS ? 2 : code == REPEAT_ZERO_3BITS ? 3 : code == REPEAT_ZERO_7BITS ? 7 : 0;
}

/* CanonicalHuffman.countSymbols :352-418 (histogram part) */
static void count_symbols(cnode_t *sym, const int32_t *text, size_t n)
{
    sym[I_END_OF_TEXT].count = 1;
    for (size_t i = 0; i < n; i++) {
        int32_t s = text[i];
        if (-128 <= s && s <= 127) sym[s + 128].count++;
        else if (-512 <= s && s <= 511) { sym[I_ESCAPE_2BITS].count++; sym[(s >> 2) + 128].count++; }
        else if (-2048 <= s && s <= 2047) { sym[I_ESCAPE_2BITS].count += 2; sym[(s >> 4) + 128].count++; }
        else if (-8192 <= s && s <= 8191) { sym[I_ESCAPE_2BITS].count += 3; sym[(s >> 6) + 128].count++; }
        else if (-32768 <= s && s <= 32767) { sym[I_ESCAPE_1BYTE].count++; sym[(s >> 8) + 128].count++; }
        else if (s == GVO_INT4_NULL) sym[I_NULL_DATA_CODE].count++;
        else if (-8388608 <= s && s <= 8388607) { sym[I_ESCAPE_1BYTE].count += 2; sym[(s >> 16) + 128].count++; }
        else { sym[I_ESCAPE_1BYTE].count += 3; sym[(s >> 24) + 128].count++; }
    }
}

/* CanonicalHuffman.encode(BitOutputStore, n, offset=0, text) :177-283 + buildCodeLengthTree :285-343.
 * Appends to the (zeroed) bit buffer at *bitPos.  The Java text loop ignores `offset` (:204) while the
 * histogram honours it (:357); every caller in the reference passes 0, and so does this restatement. */
int gvo_canon_encode(uint8_t *bits, size_t capBits, size_t *bitPos, const int32_t *text, size_t nSymbols,
                     uint8_t *codeLen260)
{
    if (nSymbols == 0 || text == NULL) return GVO_ERR_ARG;             /* IllegalArgumentException :183-185 */
    cnode_t sym[N_SYMBOLS_TOTAL];
    memset(sym, 0, sizeof sym);
    for (int i = 0; i < N_SYMBOLS_TOTAL; i++) { sym[i].isLeaf = 1; sym[i].symbol = i; }
    count_symbols(sym, text, nSymbols);
    int limited;
    int rc = build_tree(sym, N_SYMBOLS_TOTAL, &limited);
    if (rc != GVO_OK) return rc;
    int textLen[N_SYMBOLS_TOTAL];
    for (int i = 0; i < N_SYMBOLS_TOTAL; i++) {
        textLen[i] = sym[i].nBitsInCode;
        if (codeLen260) codeLen260[i] = (uint8_t)sym[i].nBitsInCode;
    }

    /* buildCodeLengthTree */
    int tCodes[N_SYMBOLS_TOTAL + 1], tRuns[N_SYMBOLS_TOTAL + 1];
    int nT = encode_lengths(N_SYMBOLS_TOTAL, textLen, tCodes, tRuns);
    cnode_t meta[SYMBOL_SET_SIZE + 1];
    memset(meta, 0, sizeof meta);
    for (int i = 0; i <= SYMBOL_SET_SIZE; i++) { meta[i].isLeaf = 1; meta[i].symbol = i; }
    meta[SYMBOL_SET_SIZE].count = 1;
    for (int i = 0; i < nT; i++) meta[tCodes[i]].count++;
    int mlimited;
    rc = build_tree(meta, SYMBOL_SET_SIZE + 1, &mlimited);
    if (rc != GVO_OK) return rc;
    int metaLen[SYMBOL_SET_SIZE + 1];
    for (int i = 0; i <= SYMBOL_SET_SIZE; i++) metaLen[i] = meta[i].nBitsInCode;
    int mCodes[SYMBOL_SET_SIZE + 2], mRuns[SYMBOL_SET_SIZE + 2];
    int nM = encode_lengths(SYMBOL_SET_SIZE + 1, metaLen, mCodes, mRuns);

    bitw_t w = {bits, capBits, *bitPos, 0};
    bw_bit(&w, 0);                                                     /* reserved bit :306 */
    for (int i = 0; i < nM; i++) {                                     /* LengthEncoder.writeEncodedLengths :169-195 */
        bw_bits(&w, 5, (uint32_t)mCodes[i]);
        int rb = run_bits(mCodes[i]);
        if (rb) bw_bits(&w, rb, (uint32_t)mRuns[i]);
    }
    for (int i = 0; i < nT; i++) {                                     /* :322-342 */
        write_symbol(&w, &meta[tCodes[i]]);
        int rb = run_bits(tCodes[i]);
        if (rb) bw_bits(&w, rb, (uint32_t)tRuns[i]);
    }

    /* the text, :203-276 */
    for (size_t i = 0; i < nSymbols; i++) {
        int32_t s = text[i];
        if (-128 <= s && s <= 127) {
            write_symbol(&w, &sym[s + 128]);
        } else if (-512 <= s && s <= 511) {
            write_symbol(&w, &sym[(s >> 2) + 128]);
            write_symbol(&w, &sym[I_ESCAPE_2BITS]); bw_bits(&w, 2, (uint32_t)(s & 3));
        } else if (-2048 <= s && s <= 2047) {
            write_symbol(&w, &sym[(s >> 4) + 128]);
            write_symbol(&w, &sym[I_ESCAPE_2BITS]); bw_bits(&w, 2, (uint32_t)((s >> 2) & 3));
            write_symbol(&w, &sym[I_ESCAPE_2BITS]); bw_bits(&w, 2, (uint32_t)(s & 3));
        } else if (-8192 <= s && s <= 8191) {
            write_symbol(&w, &sym[(s >> 6) + 128]);
            write_symbol(&w, &sym[I_ESCAPE_2BITS]); bw_bits(&w, 2, (uint32_t)((s >> 4) & 3));
            write_symbol(&w, &sym[I_ESCAPE_2BITS]); bw_bits(&w, 2, (uint32_t)((s >> 2) & 3));
            write_symbol(&w, &sym[I_ESCAPE_2BITS]); bw_bits(&w, 2, (uint32_t)(s & 3));
        } else if (-32768 <= s && s <= 32767) {
            write_symbol(&w, &sym[(s >> 8) + 128]);
            write_symbol(&w, &sym[I_ESCAPE_1BYTE]); bw_bits(&w, 8, (uint32_t)(s & 0xff));
        } else if (s == GVO_INT4_NULL) {
            write_symbol(&w, &sym[I_NULL_DATA_CODE]);
        } else if (-8333608 <= s && s <= 8388607) {                    /* sic: :258, countSymbols uses -8388608 */
            write_symbol(&w, &sym[(s >> 16) + 128]);
            write_symbol(&w, &sym[I_ESCAPE_1BYTE]); bw_bits(&w, 8, (uint32_t)((s >> 8) & 0xff));
            write_symbol(&w, &sym[I_ESCAPE_1BYTE]); bw_bits(&w, 8, (uint32_t)(s & 0xff));
        } else {
            write_symbol(&w, &sym[(s >> 24) + 128]);
            write_symbol(&w, &sym[I_ESCAPE_1BYTE]); bw_bits(&w, 8, (uint32_t)((s >> 16) & 0xff));
            write_symbol(&w, &sym[I_ESCAPE_1BYTE]); bw_bits(&w, 8, (uint32_t)((s >> 8) & 0xff));
            write_symbol(&w, &sym[I_ESCAPE_1BYTE]); bw_bits(&w, 8, (uint32_t)(s & 0xff));
        }
    }
    write_symbol(&w, &sym[I_END_OF_TEXT]);
    if (w.overflow) return GVO_ERR_CAPACITY;
    *bitPos = w.pos;
    return GVO_OK;
}

/* CanonHuffTreeDecoder(int[] lengths) :68-131 as (first code, count) per length: decoding a
 * canonical code bit by bit is equivalent to the Java lookup + node walk. */
typedef struct {
    int nUsed;
    int minLen, maxLen;
    uint32_t firstCode[17];      /* canonical code value of the first symbol of each length */
    int count[17];
    int offset[17];              /* index of that symbol in symByOrder */
    int symByOrder[MAX_ALPHABET];
} cdec_t;

static int cdec_init(cdec_t *d, const int *lengths, int n)
{
    memset(d, 0, sizeof *d);
    for (int i = 0; i < n; i++) {
        if (lengths[i] < 0 || lengths[i] > 16) return GVO_ERR_FORMAT;
        if (lengths[i] > 0) { d->count[lengths[i]]++; d->nUsed++; }
    }
    if (d->nUsed == 0) return GVO_ERR_BOUNDS;              /* sortNodes[0] on an empty array */
    int k = 0;
    for (int L = 1; L <= 16; L++) {
        d->offset[L] = k;
        for (int i = 0; i < n; i++) if (lengths[i] == L) d->symByOrder[k++] = i;
    }
    uint64_t bits = 0;
    int cur = 0, first = 1;
    for (int L = 1; L <= 16; L++) {
        if (d->count[L] == 0) continue;
        if (first) { d->minLen = L; cur = L; bits = 0; first = 0; }
        else { bits = (bits + 1) << (L - cur); cur = L; }
        d->firstCode[L] = (uint32_t)bits;
        bits += (uint64_t)(d->count[L] - 1);
        d->maxLen = L;
    }
    return GVO_OK;
}

/* one symbol; a bit pattern that is no code of an over-subscribed/incomplete table walks into a
 * missing node in Java (nodeIndex -1 -> AIOOBE): reported as GVO_ERR_BOUNDS */
static int cdec_symbol(const cdec_t *d, bitr_t *r, int *symbol)
{
    uint32_t code = 0;
    for (int L = 1; L <= d->maxLen; L++) {
        code = (code << 1) | (uint32_t)br_bit(r);
        if (r->overrun) return GVO_ERR_BOUNDS;
        if (d->count[L] && code >= d->firstCode[L] && code - d->firstCode[L] < (uint32_t)d->count[L]) {
            *symbol = d->symByOrder[d->offset[L] + (int)(code - d->firstCode[L])];
            return GVO_OK;
        }
    }
    return GVO_ERR_BOUNDS;
}

/* CanonicalHuffman.decode :441-519.  text has capacity nSymbolsInText; *nDecoded = values written. */
int gvo_canon_decode(const uint8_t *bits, size_t nBitsTotal, size_t *bitPos, int32_t *text,
                     size_t nSymbolsInText, size_t *nDecoded)
{
    if (nSymbolsInText == 0) return GVO_ERR_ARG;
    bitr_t r = {bits, nBitsTotal, *bitPos, 0};
    br_bit(&r);                                                        /* reserved bit */
    int metaLen[SYMBOL_SET_SIZE + 1 + 140];
    memset(metaLen, 0, sizeof metaLen);
    {                                                                  /* LengthEncoder.readEncodedLengths :197-236 */
        int k = 0, prior = 0, n;
        while (k < SYMBOL_SET_SIZE + 1) {
            int index = (int)br_bits(&r, 5);
            if (r.overrun) return GVO_ERR_BOUNDS;
            if (index <= MAX_STANDARD_SYMBOL) { prior = index; metaLen[k++] = index; }
            else if (index == REPEAT_PREV_2BITS) { n = (int)br_bits(&r, 2) + 3; for (int i = 0; i < n; i++) metaLen[k++] = prior; }
            else if (index == REPEAT_ZERO_3BITS) { prior = 0; n = (int)br_bits(&r, 3) + 3; for (int i = 0; i < n; i++) metaLen[k++] = 0; }
            else if (index == REPEAT_ZERO_7BITS) { prior = 0; n = (int)br_bits(&r, 7) + 11; for (int i = 0; i < n; i++) metaLen[k++] = 0; }
            if (k > SYMBOL_SET_SIZE + 1) return GVO_ERR_BOUNDS;        /* symbols[k++] past the array */
        }
    }
    cdec_t meta;
    int rc = cdec_init(&meta, metaLen, SYMBOL_SET_SIZE + 1);
    if (rc != GVO_OK) return rc;
    int textLen[N_SYMBOLS_TOTAL + 1 + 140];
    memset(textLen, 0, sizeof textLen);
    {                                                                  /* CanonHuffTreeDecoder.decodeTree :133-177 */
        int prior = 0, n;
        for (int i = 0; i < N_SYMBOLS_TOTAL; i++) {
            int test;
            rc = cdec_symbol(&meta, &r, &test);
            if (rc != GVO_OK) return rc;
            if (test <= MAX_STANDARD_SYMBOL) { textLen[i] = test; prior = test; }
            else if (test == REPEAT_PREV_2BITS) { n = (int)br_bits(&r, 2) + 3; for (int j = 0; j < n; j++) textLen[i + j] = prior; i += n - 1; }
            else if (test == REPEAT_ZERO_3BITS) { prior = 0; n = (int)br_bits(&r, 3) + 3; for (int j = 0; j < n; j++) textLen[i + j] = 0; i += n - 1; }
            else if (test == REPEAT_ZERO_7BITS) { prior = 0; n = (int)br_bits(&r, 7) + 11; for (int j = 0; j < n; j++) textLen[i + j] = 0; i += n - 1; }
            if (i >= N_SYMBOLS_TOTAL + 1) return GVO_ERR_BOUNDS;       /* wrote past int[N_SYMBOLS_TOTAL+1] */
            if (r.overrun) return GVO_ERR_BOUNDS;
        }
    }
    cdec_t tt;
    rc = cdec_init(&tt, textLen, N_SYMBOLS_TOTAL + 1);
    if (rc != GVO_OK) return rc;
    /* decodeText :469-519 */
    int32_t prior = 0;
    size_t iSymbol = 0;
    for (;;) {
        int symbol;
        rc = cdec_symbol(&tt, &r, &symbol);
        if (rc != GVO_OK) return rc;
        if (symbol == I_END_OF_TEXT) break;
        if (symbol < N_SYMBOLS_STANDARD) {
            if (iSymbol >= nSymbolsInText) return GVO_ERR_BOUNDS;
            prior = symbol - 128;
            text[iSymbol++] = prior;
        } else if (symbol == I_ESCAPE_2BITS) {
            uint32_t part = br_bits(&r, 2);
            if (iSymbol == 0) return GVO_ERR_BOUNDS;                   /* text[-1] */
            prior = (int32_t)(((uint32_t)prior << 2) | part);
            text[iSymbol - 1] = prior;
        } else if (symbol == I_ESCAPE_1BYTE) {
            uint32_t part = br_bits(&r, 8);
            if (iSymbol == 0) return GVO_ERR_BOUNDS;
            prior = (int32_t)(((uint32_t)prior << 8) | part);
            text[iSymbol - 1] = prior;
        } else if (symbol == I_NULL_DATA_CODE) {
            if (iSymbol >= nSymbolsInText) return GVO_ERR_BOUNDS;
            prior = GVO_INT4_NULL;
            text[iSymbol++] = GVO_INT4_NULL;
        }                                                              /* symbol 260 (spare slot): ignored, :512 */
        if (r.overrun) return GVO_ERR_BOUNDS;
    }
    *bitPos = r.pos;
    if (nDecoded) *nDecoded = iSymbol;
    return GVO_OK;
}

/* ---------------- integer residual streams: IPredictorModel.encodeInt / decodeInt ---------------- */

static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }

static int32_t java_floor_to_int(double x)
{
    if (x != x) return 0;
    if (x >= 2147483647.0) return 2147483647;
    if (x <= -2147483648.0) return (int32_t)0x80000000;
    return (int32_t)x;          /* x is already integral (floor applied by the caller) */
}

int gvo_predictor_encode_int(int model, int nRows, int nCols, const int32_t *v, int32_t *out, int32_t *seed)
{
    int k = 0;
    switch (model) {
    case GVO_PM_DIFFERENCING: {                 /* PredictorModelDifferencing.java:170-199 */
        *seed = v[0];
        int32_t prior = v[0];
        for (int i = 1; i < nCols; i++) { out[k++] = wsub(v[i], prior); prior = v[i]; }
        for (int r = 1; r < nRows; r++) {
            int idx = r * nCols;
            prior = v[idx - nCols];
            for (int i = 0; i < nCols; i++) { int32_t t = v[idx++]; out[k++] = wsub(t, prior); prior = t; }
        }
        return k;
    }
    case GVO_PM_LINEAR: {                       /* PredictorModelLinear.java:146-185 */
        if (nCols < 2) return -2;               /* values[1] / values[index+1]: ArrayIndexOutOfBounds */
        *seed = v[0];
        int32_t prior = v[0];
        out[k++] = wsub(v[1], prior);
        for (int r = 1; r < nRows; r++) {
            int idx = r * nCols;
            int32_t t = v[idx];
            out[k++] = wsub(t, prior);
            prior = t;
            out[k++] = wsub(v[idx + 1], prior);
        }
        for (int r = 0; r < nRows; r++) {
            int idx = r * nCols;
            int32_t a = v[idx], b = v[idx + 1];
            for (int c = 2; c < nCols; c++) {
                int32_t cv = v[idx + c];
                int32_t prediction = wsub((int32_t)(2u * (uint32_t)b), a);
                out[k++] = wsub(cv, prediction);
                a = b;
                b = cv;
            }
        }
        return k;
    }
    case GVO_PM_TRIANGLE: {                     /* PredictorModelTriangle.java:148-186 */
        if (nRows < 2 || nCols < 2) return -1;
        *seed = v[0];
        int32_t prior = v[0];
        for (int i = 1; i < nCols; i++) { out[k++] = wsub(v[i], prior); prior = v[i]; }
        prior = v[0];
        for (int i = 1; i < nRows; i++) { int32_t t = v[i * nCols]; out[k++] = wsub(t, prior); prior = t; }
        for (int r = 1; r < nRows; r++) {
            int k1 = r * nCols, k0 = k1 - nCols;
            for (int i = 1; i < nCols; i++) {
                int32_t za = v[k0++], zb = v[k1++], zc = v[k0];
                int32_t prediction = wsub(wadd(zc, zb), za);
                out[k++] = wsub(v[k1], prediction);
            }
        }
        return k;
    }
    case GVO_PM_DIFFERENCING_NULLS: {           /* PredictorModelDifferencingWithNulls.java:169-237 */
        int64_t sumStart = 0;
        int64_t nStart = 0;
        int nullFlag = 1;
        for (int r = 0; r < nRows; r++) {
            int ro = r * nCols;
            for (int c = 0; c < nCols; c++) {
                int32_t t = v[ro + c];
                if (t == GVO_INT4_NULL) nullFlag = 1;
                else { if (nullFlag) { sumStart += t; nStart++; } nullFlag = 0; }
            }
            nullFlag = v[ro] == GVO_INT4_NULL;
        }
        if (nStart == 0) return 0;
        double avg = (double)sumStart / (double)nStart;
        double f = avg + 0.5;
        double fl = (double)(int64_t)f;
        if (fl > f) fl -= 1.0;
        int32_t es = java_floor_to_int(fl);
        *seed = es;
        int64_t prior = es;
        nullFlag = 0;
        for (int r = 0; r < nRows; r++) {
            int idx = r * nCols;
            for (int c = 0; c < nCols; c++) {
                int32_t t = v[idx++];
                if (t == GVO_INT4_NULL) { nullFlag = 1; out[k++] = GVO_INT4_NULL; }
                else {
                    if (nullFlag) { prior = es; nullFlag = 0; }
                    out[k++] = (int32_t)(uint32_t)((uint64_t)(int64_t)t - (uint64_t)prior);
                    prior = t;
                }
            }
            prior = v[r * nCols];
            nullFlag = v[r * nCols] == GVO_INT4_NULL;
        }
        return k;
    }
    default: return -3;
    }
}

int gvo_predictor_decode_int(int model, int32_t seed, int nRows, int nCols, const int32_t *e, int32_t *o)
{
    int k = 0;
    switch (model) {
    case GVO_PM_DIFFERENCING: {                 /* PredictorModelDifferencing.java:203-224 */
        o[0] = seed;
        int32_t prior = seed;
        for (int i = 1; i < nCols; i++) { prior = wadd(prior, e[k++]); o[i] = prior; }
        for (int r = 1; r < nRows; r++) {
            int idx = r * nCols;
            prior = o[idx - nCols];
            for (int c = 0; c < nCols; c++) { prior = wadd(prior, e[k++]); o[idx++] = prior; }
        }
        return GVO_OK;
    }
    case GVO_PM_LINEAR: {                       /* PredictorModelLinear.java:188-222 */
        if (nCols < 2) return GVO_ERR_BOUNDS;
        int32_t prior = seed;
        o[0] = seed;
        o[1] = wadd(e[k++], prior);
        for (int r = 1; r < nRows; r++) {
            int idx = r * nCols;
            int32_t t = wadd(e[k++], prior);
            o[idx] = t;
            prior = t;
            o[idx + 1] = wadd(e[k++], t);
        }
        for (int r = 0; r < nRows; r++) {
            int idx = r * nCols;
            int32_t a = o[idx], b = o[idx + 1];
            for (int c = 2; c < nCols; c++) {
                int32_t prediction = wsub((int32_t)(2u * (uint32_t)b), a);
                int32_t cv = wadd(prediction, e[k++]);
                a = b;
                b = cv;
                o[idx + c] = cv;
            }
        }
        return GVO_OK;
    }
    case GVO_PM_TRIANGLE: {                     /* PredictorModelTriangle.java:189-215 */
        o[0] = seed;
        int32_t prior = seed;
        for (int i = 1; i < nCols; i++) { prior = wadd(prior, e[k++]); o[i] = prior; }
        prior = seed;
        for (int i = 1; i < nRows; i++) { prior = wadd(prior, e[k++]); o[i * nCols] = prior; }
        for (int r = 1; r < nRows; r++) {
            int k1 = r * nCols, k0 = k1 - nCols;
            for (int i = 1; i < nCols; i++) {
                int32_t za = o[k0++], zb = o[k1++], zc = o[k0];
                o[k1] = wadd(wsub(wadd(zb, zc), za), e[k++]);
            }
        }
        return GVO_OK;
    }
    case GVO_PM_DIFFERENCING_NULLS: {           /* PredictorModelDifferencingWithNulls.java:240-268 */
        int32_t prior = seed;
        int nullFlag = 1;
        for (int r = 0; r < nRows; r++) {
            int idx = r * nCols;
            for (int c = 0; c < nCols; c++) {
                int32_t t = e[k++];
                if (t == GVO_INT4_NULL) { nullFlag = 1; o[idx++] = GVO_INT4_NULL; }
                else {
                    if (nullFlag) { nullFlag = 0; prior = seed; }
                    prior = wadd(prior, t);
                    o[idx++] = prior;
                }
            }
            prior = o[r * nCols];
            nullFlag = prior == GVO_INT4_NULL;
        }
        return GVO_OK;
    }
    default: return GVO_ERR_FORMAT;
    }
}

/* ---------------- CodecCanonHuffman (compress/canonicalHuffman/CodecCanonHuffman.java) ---------------- */

size_t gvo_codec_canon_bound(size_t nCells)
{
    /* 6 header bytes + code tables (< 1 KB) + at most 4 symbols of <= 15 bits and 24 raw bits per value + EOT */
    return 6 + 1024 + (nCells * 84 + 15 + 7) / 8 + 8;
}

/* encode :70-143, compress :145-160.  predictorMask as gvo_codec_huffman_encode. */
int gvo_codec_canon_encode(int codecIndex, int nRows, int nCols, const int32_t *values, uint8_t *out,
                           size_t outCap, size_t *outLen, int predictorMask, int *predictorUsed)
{
    const size_t nCells = (size_t)nRows * (size_t)nCols;
    if (nRows <= 0 || nCols <= 0) return GVO_ERR_ARG;
    int hasNull = 0, hasValid = 0;
    for (size_t i = 0; i < nCells; i++) {
        if (values[i] == GVO_INT4_NULL) hasNull = 1;
        else hasValid = 1;
    }
    if (predictorUsed) *predictorUsed = 0;
    if (!hasValid) return GVO_DECLINED;
    int uniform = 1;
    for (size_t i = 1; i < nCells; i++) if (values[i] != values[0]) { uniform = 0; break; }
    if (uniform) {                                                     /* :95-110 */
        if (outCap < 6) return GVO_ERR_CAPACITY;
        out[0] = (uint8_t)codecIndex;
        out[1] = 0;
        uint32_t s = (uint32_t)values[0];
        out[2] = (uint8_t)s; out[3] = (uint8_t)(s >> 8); out[4] = (uint8_t)(s >> 16); out[5] = (uint8_t)(s >> 24);
        *outLen = 6;
        return GVO_OK;
    }
    const size_t cap = gvo_codec_canon_bound(nCells);
    uint8_t *best = NULL, *cand = malloc(cap);
    int32_t *res = malloc(sizeof(int32_t) * nCells);
    size_t bestLen = (size_t)-1;
    int rc = GVO_OK, bestModel = 0;
    for (int model = 1; model <= 4 && rc == GVO_OK; model++) {
        if (hasNull != (model == GVO_PM_DIFFERENCING_NULLS)) continue;
        if (!((predictorMask >> (model - 1)) & 1)) continue;
        int32_t seed = 0;
        int n = gvo_predictor_encode_int(model, nRows, nCols, values, res, &seed);
        if (n == -2) { rc = GVO_ERR_BOUNDS; break; }                   /* Linear on a 1-column tile */
        if (n <= 0) { rc = GVO_ERR_ARG; break; }                       /* CanonicalHuffman.encode :183 throws */
        memset(cand, 0, cap);
        cand[0] = (uint8_t)codecIndex;
        cand[1] = (uint8_t)model;
        cand[2] = (uint8_t)seed; cand[3] = (uint8_t)((uint32_t)seed >> 8);
        cand[4] = (uint8_t)((uint32_t)seed >> 16); cand[5] = (uint8_t)((uint32_t)seed >> 24);
        size_t bitPos = 48;
        rc = gvo_canon_encode(cand, cap * 8, &bitPos, res, (size_t)n, NULL);
        if (rc != GVO_OK) break;
        size_t len = (bitPos + 7) / 8;
        if (len < bestLen) {                                           /* strict: :133 */
            bestLen = len;
            bestModel = model;
            uint8_t *t = best; best = cand; cand = t ? t : malloc(cap);
        }
    }
    if (rc == GVO_OK) {
        if (best == NULL) rc = GVO_DECLINED;
        else if (bestLen > outCap) rc = GVO_ERR_CAPACITY;
        else { memcpy(out, best, bestLen); *outLen = bestLen; if (predictorUsed) *predictorUsed = bestModel; }
    }
    free(best); free(cand); free(res);
    return rc;
}

/* decode :163-195 */
int gvo_codec_canon_decode(int nRows, int nCols, const uint8_t *packing, size_t len, int32_t *values)
{
    const size_t nCells = (size_t)nRows * (size_t)nCols;
    if (len < 6) return GVO_ERR_BOUNDS;
    int predictor = (int8_t)packing[1];
    int32_t seed = (int32_t)((uint32_t)packing[2] | ((uint32_t)packing[3] << 8) | ((uint32_t)packing[4] << 16) |
                             ((uint32_t)packing[5] << 24));
    if (predictor == 0 && len == 6) {
        for (size_t i = 0; i < nCells; i++) values[i] = seed;
        return GVO_OK;
    }
    if (predictor < 1 || predictor > 4) return GVO_ERR_FORMAT;         /* IOException :208-209 */
    int32_t *res = calloc(nCells ? nCells : 1, sizeof(int32_t));
    size_t bitPos = 48, nDec = 0;
    int rc = gvo_canon_decode(packing, len * 8, &bitPos, res, nCells, &nDec);
    if (rc == GVO_OK) rc = gvo_predictor_decode_int(predictor, seed, nRows, nCols, res, values);
    free(res);
    return rc;
}

int gvo_batch_canon_encode(int codecIndex, int nRows, int nCols, size_t nTiles, const int32_t *values,
                           uint8_t *out, size_t stride, uint32_t *lengths, uint8_t *predictors)
{
    const size_t nCells = (size_t)nRows * (size_t)nCols;
    for (size_t t = 0; t < nTiles; t++) {
        size_t len = 0;
        int pu = 0;
        int rc = gvo_codec_canon_encode(codecIndex, nRows, nCols, values + t * nCells, out + t * stride, stride,
                                        &len, 0xF, &pu);
        if (rc < 0) return rc;
        lengths[t] = rc == GVO_OK ? (uint32_t)len : 0;
        if (predictors) predictors[t] = (uint8_t)pu;
    }
    return GVO_OK;
}

int gvo_batch_canon_decode(int nRows, int nCols, size_t nTiles, const uint8_t *packings, size_t stride,
                           const uint32_t *lengths, int32_t *values)
{
    const size_t nCells = (size_t)nRows * (size_t)nCols;
    for (size_t t = 0; t < nTiles; t++) {
        int rc = gvo_codec_canon_decode(nRows, nCols, packings + t * stride, lengths[t], values + t * nCells);
        if (rc != GVO_OK) return rc;
    }
    return GVO_OK;
}
