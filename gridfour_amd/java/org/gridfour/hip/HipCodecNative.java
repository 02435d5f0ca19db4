/*
 * Native entry points shared by the adapters of this package (JNI shim: gvrs_hip_jni.cpp; C ABI:
 * include/gvrs_hip_codec.h).  kind: 0 CodecHuffman, 1 CodecCanonHuffman, 2 LSOP12 without the Deflate
 * alternative, 3 LSOP12 with it (the reference's default), 4 CodecDeflate.  Not compiled in the build image (no JDK).
 */
package org.gridfour.hip;

import java.io.IOException;

final class HipCodecNative {
  static {
    System.loadLibrary("gvrs_hip_jni");
  }

  static native long create(int device);
  static native void destroy(long handle);
  static native byte[] encode(long handle, int kind, int codecIndex, int nRows, int nCols, int[] values);
  static native int[] decode(long handle, int kind, int nRows, int nColumns, byte[] packing) throws IOException;

  /** CodecFloat.encodeFloats / decodeFloats (gf_float_encode_f32 / gf_float_decode_f32); level: zlib's, 9 in the current source. */
  static native byte[] encodeFloats(long handle, int codecIndex, int nRows, int nCols, float[] values, int level);
  static native float[] decodeFloats(long handle, int nRows, int nColumns, byte[] packing) throws IOException;

  /**
   * All dirty tiles of a flush in one call (RecordManager.writeTile framing, gf_tile_record_encode_batch):
   * codecKinds lists the file's codecs in CodecMaster order (GF_CODEC_* of the C header; empty when compression
   * is off); cells holds nTiles x nRows x nCols values (int[] or short[], elemType 0 / 1); recordOffsets receives
   * nTiles + 1 offsets into the returned bytes, which are the finished tile records, ready to append to the file.
   */
  static native byte[] tileRecords(long handle, int[] codecKinds, int elemType, int fillValue, int nRows, int nCols,
    int[] tileIndices, Object cells, boolean checksums, long[] recordOffsets) throws IOException;

  /**
   * The read side (RecordManager.readTile for a batch, gf_tile_record_decode_batch): records back to tile indices
   * and cells; status[t] != 0 marks a record that the reference would have rejected with an IOException.
   */
  static native void tilesFromRecords(long handle, int[] codecKinds, int elemType, int nRows, int nCols, byte[] records,
    long[] recordOffsets, boolean verifyChecksums, int[] tileIndices, Object cells, int[] status) throws IOException;

  // ---- read-ahead (gf_readahead_*): gvrs/TileDecompressionAssistant.java as an N-tile prefetch queue ----
  static native long readaheadCreate(int device, int[] codecKinds, int nRows, int nCols, int maxBatch);
  static native void readaheadDestroy(long ra);
  static native void readaheadSubmit(long ra, int tileIndex, byte[] packing);
  static native int readaheadPending(long ra);
  /**
   * Waits while waitIndex is queued or being decoded, then hands over up to indices.length finished tiles: returns
   * their number n, fills indices[0..n), status[0..n) and cells[i * nRows * nCols ...] (the tile waited for comes first).
   */
  static native int readaheadTake(long ra, int waitIndex, int[] indices, int[] cells, int[] status);

  // ---- several GPUs from one JVM (gf_multi_*): contiguous tile ranges per device, no exchange between devices ----
  static native long multiCreate(int[] devices);
  static native void multiDestroy(long multi);
  /** CodecHuffman.encode of nTiles tiles (cells: nTiles x nRows x nCols) over all devices; offsets receives nTiles + 1. */
  static native byte[] multiHuffmanEncode(long multi, int codecIndex, int nRows, int nCols, int[] cells, long[] offsets,
    byte[] predictors, int[] status) throws IOException;
  static native void multiHuffmanDecode(long multi, int nRows, int nCols, byte[] blob, long[] offsets, int[] cells,
    int[] status) throws IOException;

  private HipCodecNative() {
  }
}
