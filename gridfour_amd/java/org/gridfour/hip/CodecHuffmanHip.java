/*
 * Adapter that registers the MI355X codec under Gridfour's plug-in interface.
 *
 *   GvrsFileSpecification spec = ...;
 *   spec.setDataCompressionEnabled(true);
 *   spec.addCompressionCodec("GvrsHuffman", org.gridfour.hip.CodecHuffmanHip.class);
 *
 * Re-using the id "GvrsHuffman" replaces the decoder-only default entry
 * (gvrs/GvrsFileSpecification.java:221-226, 1576-1631), so files written through this adapter are
 * readable by stock Gridfour (the packings are byte-identical to CodecHuffman's).  The class
 * implements both interfaces directly, as addCompressionCodec requires (getInterfaces() test,
 * GvrsFileSpecification.java:1608-1626), and has the public no-argument constructor CodecHolder
 * needs (gvrs/CodecHolder.java:189-206).
 *
 * Not compiled in the build image (no JDK); see INTEGRATION.md.
 */
package org.gridfour.hip;

import java.io.IOException;
import java.io.PrintStream;
import org.gridfour.compress.CodecHuffman;
import org.gridfour.compress.ICompressionDecoder;
import org.gridfour.compress.ICompressionEncoder;

public class CodecHuffmanHip implements ICompressionEncoder, ICompressionDecoder {

  static {
    System.loadLibrary("gvrs_hip_jni");
  }

  private static native long createNative(int device);
  private static native void destroyNative(long handle);
  private static native byte[] encodeNative(long handle, int codecIndex, int nRows, int nCols, int[] values);
  private static native int[] decodeNative(long handle, int nRows, int nColumns, byte[] packing) throws IOException;

  private final long handle;
  /** analysis statistics are host-side bookkeeping: delegate to the stock implementation */
  private final CodecHuffman statsDelegate = new CodecHuffman();

  public CodecHuffmanHip() {
    handle = createNative(Integer.getInteger("gridfour.hip.device", 0));
  }

  @Override
  public byte[] encode(int codecIndex, int nRows, int nCols, int[] values) {
    return encodeNative(handle, codecIndex, nRows, nCols, values);
  }

  @Override
  public int[] decode(int nRows, int nColumns, byte[] packing) throws IOException {
    return decodeNative(handle, nRows, nColumns, packing);
  }

  @Override
  public byte[] encodeFloats(int codecIndex, int nRows, int nCols, float[] values) {
    return null;
  }

  @Override
  public float[] decodeFloats(int nRows, int nColumns, byte[] packing) throws IOException {
    return null;
  }

  @Override
  public boolean implementsFloatingPointEncoding() {
    return false;
  }

  @Override
  public boolean implementsIntegerEncoding() {
    return true;
  }

  @Override
  public void analyze(int nRows, int nColumns, byte[] packing) throws IOException {
    statsDelegate.analyze(nRows, nColumns, packing);
  }

  @Override
  public void reportAnalysisData(PrintStream ps, int nTilesInRaster) {
    statsDelegate.reportAnalysisData(ps, nTilesInRaster);
  }

  @Override
  public void clearAnalysisData() {
    statsDelegate.clearAnalysisData();
  }

  @Override
  protected void finalize() {
    destroyNative(handle);
  }
}
