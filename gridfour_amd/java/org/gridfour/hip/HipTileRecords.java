/*
 * The flush-side and read-side batch binding (SURVEY 8 f3): all dirty tiles of GvrsFile.flush()
 * (gvrs/RasterTileCache.java:286-294 -> gvrs/RecordManager.java:386-490, one writeTile per tile in the reference) go
 * to the GPU in ONE call and come back as finished tile records -- record header, tile index, element length, packing
 * or standard form, zero padding, CRC-32C -- exactly as RecordManager.writeTile + fileSpaceAlloc + fileSpaceFinishRecord
 * lay them out (gvrs/RecordManager.java:153-204, 217-262).  The caller appends the bytes and sets one file position per
 * tile.  RecordManager and TileDirectory are package-private, so the three lines that use this class live in
 * org.gridfour.gvrs (INTEGRATION.md section 2 shows them); this class needs nothing from that package.
 * Not compiled in the build image (no JDK).
 */
package org.gridfour.hip;

import java.io.IOException;

public final class HipTileRecords implements AutoCloseable {

  /** element types of gf_tile_record_* */
  public static final int ELEM_INT = 0;
  public static final int ELEM_SHORT = 1;
  /** codec kinds in CodecMaster order: GF_CODEC_* of include/gvrs_hip_codec.h */
  public static final int CODEC_NONE = 0;
  public static final int CODEC_HUFFMAN = 1;
  public static final int CODEC_DEFLATE = 2;
  public static final int CODEC_CANON_HUFFMAN = 3;
  public static final int CODEC_LSOP12 = 4;
  /** bytes between the start of a record and its tile content (int32 size, type, 3 pad): RecordManager.RECORD_HEADER_SIZE */
  public static final int RECORD_HEADER_SIZE = 8;

  private long handle;
  private final int[] codecKinds;
  private final int nRows;
  private final int nCols;

  /**
   * @param device the GPU
   * @param codecKinds the file's codec list (GvrsFileSpecification.getCompressionCodecs order); empty = no compression
   */
  public HipTileRecords(int device, int[] codecKinds, int nRowsInTile, int nColsInTile) {
    this.handle = HipCodecNative.create(device);
    this.codecKinds = codecKinds.clone();
    this.nRows = nRowsInTile;
    this.nCols = nColsInTile;
  }

  /**
   * The records of a batch of dirty tiles of one integer element.
   *
   * @param tileIndices the tiles, in the order their cells are given
   * @param cells tileIndices.length x nRows x nCols values
   * @param recordOffsets receives tileIndices.length + 1 offsets into the returned bytes; record t starts at
   * recordOffsets[t], its tile content at recordOffsets[t] + RECORD_HEADER_SIZE (what TileDirectory stores)
   * @return the records, back to back, ready for one writeFully at the end of the file
   */
  public synchronized byte[] encode(int[] tileIndices, int[] cells, boolean checksums, long[] recordOffsets)
    throws IOException {
    check(tileIndices.length, cells.length, recordOffsets.length);
    return HipCodecNative.tileRecords(handle, codecKinds, ELEM_INT, 0, nRows, nCols, tileIndices, cells, checksums,
      recordOffsets);
  }

  /** The same for a short element; fillValue is the element's fill (it travels as the null code through the codecs). */
  public synchronized byte[] encode(int[] tileIndices, short[] cells, int fillValue, boolean checksums,
    long[] recordOffsets) throws IOException {
    check(tileIndices.length, cells.length, recordOffsets.length);
    return HipCodecNative.tileRecords(handle, codecKinds, ELEM_SHORT, fillValue, nRows, nCols, tileIndices, cells,
      checksums, recordOffsets);
  }

  /**
   * The read side (RecordManager.readTile :472-520 for a batch, e.g. everything a read-ahead fetched).
   *
   * @param records the records, back to back, with recordOffsets (nTiles + 1 entries) as produced above or taken from
   * the tile directory
   * @param tileIndices receives the tile index stored in every record
   * @param cells receives nTiles x nRows x nCols values
   * @param status receives 0 per record, or the (negative) status of a record the reference rejects with an IOException
   */
  public synchronized void decode(byte[] records, long[] recordOffsets, boolean verifyChecksums, int[] tileIndices,
    int[] cells, int[] status) throws IOException {
    check(tileIndices.length, cells.length, recordOffsets.length);
    HipCodecNative.tilesFromRecords(handle, codecKinds, ELEM_INT, nRows, nCols, records, recordOffsets, verifyChecksums,
      tileIndices, cells, status);
  }

  private void check(int nTiles, int nCells, int nOffsets) {
    if (handle == 0) {
      throw new IllegalStateException("closed");
    }
    if ((long) nTiles * nRows * nCols != nCells || nOffsets < nTiles + 1) {
      throw new IllegalArgumentException("cells must hold nTiles x nRows x nCols values, offsets nTiles + 1 entries");
    }
  }

  @Override
  public synchronized void close() {
    if (handle != 0) {
      HipCodecNative.destroy(handle);
      handle = 0;
    }
  }
}
