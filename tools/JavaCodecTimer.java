/*
 * JavaCodecTimer -- times the REFERENCE Java codec (org.gridfour.compress.CodecHuffman and its siblings, from a Gridfour jar
 * on the class path) on the tiles bench.py hands it, so that the bench line can carry the north star's baseline: "the
 * reference Java CPU codec timed on the same box's host cores" (BASELINE.md section 3).  bench.py runs it only where a
 * `java` launcher (11+: source-file mode, no javac needed) AND a Gridfour jar exist; the build image has neither, so this file
 * has never been compiled there -- it uses nothing beyond java.base and the four codec classes' public encode / decode.
 *
 *   java -cp <gridfour-core.jar> tools/JavaCodecTimer.java <tiles.raw> <nRows> <nCols> <nTiles> <codec> <packings.out>
 *
 * tiles.raw: nTiles x nRows x nCols little-endian int32.  codec: huffman | canon | lsop.  Single thread.  Warm-up: 20 passes
 * over the first min(nTiles, 64) tiles (JIT), then the whole sample five times, best pass of each direction reported.
 * packings.out: per tile a little-endian int32 length (-1: the encoder declined) and the packing's bytes -- bench.py compares
 * them with the GPU's packings byte for byte: the metric's "bit-exact vs Java ref", checked against the real thing.
 * Prints one JSON line.
 */
import java.io.BufferedOutputStream;
import java.io.DataOutputStream;
import java.io.FileOutputStream;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.IntBuffer;
import java.nio.file.Files;
import java.nio.file.Paths;
import java.util.Arrays;
import org.gridfour.compress.ICompressionDecoder;
import org.gridfour.compress.ICompressionEncoder;

public class JavaCodecTimer {

  public static void main(String[] args) throws Exception {
    if (args.length < 6) {
      System.err.println("usage: JavaCodecTimer tiles.raw nRows nCols nTiles huffman|canon|lsop packings.out");
      System.exit(2);
    }
    final int nRows = Integer.parseInt(args[1]), nCols = Integer.parseInt(args[2]), nTiles = Integer.parseInt(args[3]);
    final String codec = args[4];
    final int cells = nRows * nCols;
    byte[] raw = Files.readAllBytes(Paths.get(args[0]));
    if (raw.length < (long) nTiles * cells * 4) {
      throw new IllegalArgumentException("tiles.raw holds fewer than nTiles tiles");
    }
    IntBuffer ib = ByteBuffer.wrap(raw).order(ByteOrder.LITTLE_ENDIAN).asIntBuffer();
    int[][] tiles = new int[nTiles][cells];
    for (int t = 0; t < nTiles; t++) {
      ib.get(tiles[t]);
    }
    raw = null;

    ICompressionEncoder enc;
    ICompressionDecoder dec;
    if (codec.equals("huffman")) {
      org.gridfour.compress.CodecHuffman c = new org.gridfour.compress.CodecHuffman();
      enc = c;
      dec = c;
    } else if (codec.equals("canon")) {
      org.gridfour.compress.canonicalHuffman.CodecCanonHuffman c = new org.gridfour.compress.canonicalHuffman.CodecCanonHuffman();
      enc = c;
      dec = c;
    } else if (codec.equals("lsop")) {
      org.gridfour.lsop.LsEncoder12 e = new org.gridfour.lsop.LsEncoder12();
      e.setDeflateEnabled(false);          // the container the device-resident bench line times (canonical Huffman)
      enc = e;
      dec = new org.gridfour.lsop.LsDecoder12();
    } else {
      throw new IllegalArgumentException("codec: huffman, canon or lsop");
    }

    final int nWarm = Math.min(nTiles, 64);
    byte[][] packs = new byte[nTiles][];
    long sink = 0;
    for (int pass = 0; pass < 20; pass++) {
      for (int t = 0; t < nWarm; t++) {
        byte[] p = enc.encode(0, nRows, nCols, tiles[t]);
        if (p != null) {
          sink += dec.decode(nRows, nCols, p)[cells - 1];
        }
      }
    }
    long bestEnc = Long.MAX_VALUE, bestDec = Long.MAX_VALUE;
    boolean exact = true;
    for (int pass = 0; pass < 5; pass++) {
      long t0 = System.nanoTime();
      for (int t = 0; t < nTiles; t++) {
        packs[t] = enc.encode(0, nRows, nCols, tiles[t]);
      }
      long t1 = System.nanoTime();
      for (int t = 0; t < nTiles; t++) {
        if (packs[t] != null) {
          int[] back = dec.decode(nRows, nCols, packs[t]);
          sink += back[0];
          if (pass == 0 && !Arrays.equals(back, tiles[t])) {
            exact = false;
          }
        }
      }
      long t2 = System.nanoTime();
      bestEnc = Math.min(bestEnc, t1 - t0);
      bestDec = Math.min(bestDec, t2 - t1);
    }
    long packedBytes = 0;
    int declined = 0;
    try (DataOutputStream out = new DataOutputStream(new BufferedOutputStream(new FileOutputStream(args[5])))) {
      byte[] len = new byte[4];
      for (int t = 0; t < nTiles; t++) {
        int n = packs[t] == null ? -1 : packs[t].length;
        ByteBuffer.wrap(len).order(ByteOrder.LITTLE_ENDIAN).putInt(n);
        out.write(len);
        if (n > 0) {
          out.write(packs[t]);
          packedBytes += n;
        } else if (n < 0) {
          declined++;
        }
      }
    }
    final double mb = (double) nTiles * cells * 4 / 1e6;
    final double encS = bestEnc / 1e9, decS = bestDec / 1e9;
    System.out.printf(java.util.Locale.ROOT,
      "{\"codec\": \"%s\", \"tiles\": %d, \"mb\": %.3f, \"encode_MBps\": %.3f, \"decode_MBps\": %.3f, \"roundtrip_MBps\": %.3f, "
      + "\"threads\": 1, \"warmup_passes\": 20, \"timed_passes\": 5, \"java_roundtrip_exact\": %b, \"declined\": %d, "
      + "\"packed_bytes\": %d, \"jvm\": \"%s %s\", \"sink\": %d}%n",
      codec, nTiles, mb, mb / encS, mb / decS, mb / (encS + decS), exact, declined, packedBytes,
      System.getProperty("java.vm.name"), System.getProperty("java.version"), sink & 1);
  }
}
