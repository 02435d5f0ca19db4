/*
 * Adapter that registers the MI355X implementation of CodecDeflate (standard codec "GvrsDeflate",
 * gvrs/GvrsFileSpecification.java:227) under Gridfour's plug-in interface.
 *
 *   spec.addCompressionCodec("GvrsDeflate", org.gridfour.hip.CodecDeflateHip.class);
 *
 * The packings are byte-identical to CodecDeflate's, so files stay readable by stock Gridfour
 * (the class name is persisted in the file's GvrsJavaCodecs metadata and falls back to the standard
 * class when this one is missing, gvrs/GvrsFileSpecification.java:303-310).  Public no-argument
 * constructor and both interfaces in the class's own implements clause, as CodecHolder and
 * addCompressionCodec require.  Not compiled in the build image (no JDK); see INTEGRATION.md.
 */
package org.gridfour.hip;

import java.io.IOException;
import java.io.PrintStream;
import org.gridfour.compress.ICompressionDecoder;
import org.gridfour.compress.ICompressionEncoder;
import org.gridfour.compress.CodecDeflate;

public class CodecDeflateHip implements ICompressionEncoder, ICompressionDecoder {

  private static final int KIND = 4;
  private final long handle = HipCodecNative.create(Integer.getInteger("gridfour.hip.device", 0));
  /** analysis statistics are host-side bookkeeping: delegate to the stock implementation */
  private final CodecDeflate statsDelegate = new CodecDeflate();

  public CodecDeflateHip() {
  }

  @Override
  public byte[] encode(int codecIndex, int nRows, int nCols, int[] values) {
    return HipCodecNative.encode(handle, KIND, codecIndex, nRows, nCols, values);
  }

  @Override
  public int[] decode(int nRows, int nColumns, byte[] packing) throws IOException {
    return HipCodecNative.decode(handle, KIND, nRows, nColumns, packing);
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
    HipCodecNative.destroy(handle);
  }
}
