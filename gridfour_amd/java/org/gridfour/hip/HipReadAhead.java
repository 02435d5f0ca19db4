/*
 * The tile cache's reading assistant (gvrs/TileDecompressionAssistant.java:60-230) on the GPU, generalised from one
 * predicted tile to a window of them: everything that is queued when the native worker wakes up is decoded as ONE GPU
 * batch (gf_readahead_*).  Same three operations as the reference's class, under its names:
 *
 *   submitDecompression(tileIndex, packing)    TileDecompressionAssistant.submitDecompression :103-109
 *   getPendingTaskCount()                      :230
 *   getTilesWithWaitForIndex(targetIndex, ...) :176-214
 *
 * and RasterTileCache.readTileUsingAssistant (gvrs/RasterTileCache.java:339-426) keeps its shape; only its
 * "pending < 2, predict index + 1" becomes "pending < window, predict the next tiles of the row".  The packing is the
 * element's bytes as RecordManager.readTilePacking returns them (one integer element per tile).
 * Not compiled in the build image (no JDK).
 */
package org.gridfour.hip;

public final class HipReadAhead implements AutoCloseable {

  private long ra;
  private final int cellsPerTile;

  public HipReadAhead(int device, int[] codecKinds, int nRowsInTile, int nColsInTile, int maxTilesPerBatch) {
    this.ra = HipCodecNative.readaheadCreate(device, codecKinds, nRowsInTile, nColsInTile, maxTilesPerBatch);
    this.cellsPerTile = nRowsInTile * nColsInTile;
  }

  /** queues a tile for background decoding; the packing is copied, the call returns at once */
  public void submitDecompression(int tileIndex, byte[] packing) {
    HipCodecNative.readaheadSubmit(handle(), tileIndex, packing);
  }

  /** tiles queued or being decoded */
  public int getPendingTaskCount() {
    return HipCodecNative.readaheadPending(handle());
  }

  /**
   * Waits while targetIndex is queued or being decoded, then hands over finished tiles (the target first).
   *
   * @param indices receives the tile indices; its length bounds the number of tiles taken
   * @param cells receives the values, tile i at i * nRows * nCols; exactly indices.length tiles long
   * @param status receives 0, or the status of a tile the reference's decoder would have rejected
   * @return the number of tiles handed over (0: nothing was finished and the target was never submitted)
   */
  public int getTilesWithWaitForIndex(int targetIndex, int[] indices, int[] cells, int[] status) {
    if ((long) indices.length * cellsPerTile != cells.length || status.length < indices.length) {
      throw new IllegalArgumentException("cells must hold exactly indices.length tiles, status at least as many entries");
    }
    return HipCodecNative.readaheadTake(handle(), targetIndex, indices, cells, status);
  }

  private long handle() {
    if (ra == 0) {
      throw new IllegalStateException("closed");
    }
    return ra;
  }

  @Override
  public synchronized void close() {
    if (ra != 0) {
      HipCodecNative.readaheadDestroy(ra);
      ra = 0;
    }
  }
}
