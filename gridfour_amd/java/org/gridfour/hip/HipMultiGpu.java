/*
 * All GPUs of a node from one JVM (gf_multi_*): a batch of tiles is cut into contiguous tile ranges, one per device,
 * with no exchange between devices (tiles are independent, gvrs/RasterTile.java:237-241); packings come back
 * concatenated in tile order, byte for byte what one device gives.  This is what the tile loops of
 * gvrs/CodecMaster.java:142-203 become when a flush is handed over as a whole.
 * Not compiled in the build image (no JDK).
 */
package org.gridfour.hip;

import java.io.IOException;

public final class HipMultiGpu implements AutoCloseable {

  private long multi;

  /** @param devices the GPUs to use (a device may be listed more than once) */
  public HipMultiGpu(int[] devices) {
    this.multi = HipCodecNative.multiCreate(devices);
  }

  /**
   * CodecHuffman.encode of every tile (all predictors tried, the shortest packing kept).
   *
   * @param cells nTiles x nRows x nCols values
   * @param offsets receives nTiles + 1 offsets into the returned bytes (an empty range = the encoder returned null)
   * @param predictors receives the predictor code of every packing (may be null)
   * @param status receives 0 / 1 (null) per tile (may be null)
   */
  public synchronized byte[] encodeHuffman(int codecIndex, int nRows, int nCols, int[] cells, long[] offsets,
    byte[] predictors, int[] status) throws IOException {
    return HipCodecNative.multiHuffmanEncode(handle(), codecIndex, nRows, nCols, cells, offsets, predictors, status);
  }

  /** CodecHuffman.decode of every packing; status[t] != 0 marks a packing the reference rejects with an IOException */
  public synchronized void decodeHuffman(int nRows, int nCols, byte[] blob, long[] offsets, int[] cells, int[] status)
    throws IOException {
    HipCodecNative.multiHuffmanDecode(handle(), nRows, nCols, blob, offsets, cells, status);
  }

  private long handle() {
    if (multi == 0) {
      throw new IllegalStateException("closed");
    }
    return multi;
  }

  @Override
  public synchronized void close() {
    if (multi != 0) {
      HipCodecNative.multiDestroy(multi);
      multi = 0;
    }
  }
}
