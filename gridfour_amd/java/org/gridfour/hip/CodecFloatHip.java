/*
 * Adapter that registers the MI355X implementation of CodecFloat (standard codec "GvrsFloat",
 * gvrs/GvrsFileSpecification.java:227-229) under Gridfour's plug-in interface.
 *
 *   spec.addCompressionCodec("GvrsFloat", org.gridfour.hip.CodecFloatHip.class);
 *
 * The five byte planes of a float tile (sign bits, exponent, three byte-delta coded mantissa bytes,
 * compress/CodecFloat.java:328-369) are split and merged on the GPU; each plane's Deflate stream is zlib's
 * (level 9 in the current source, CodecFloat.java:268-283; -Dgridfour.hip.floatLevel=6 reproduces the
 * reference's sample files) and is inflated on the GPU on the way back.  The packings are byte-identical
 * to CodecFloat's where zlib's are.  As CodecFloat itself (:116-125, :461-468) the integer halves of the
 * two interfaces refuse: this codec implements floating-point encoding only.  Public no-argument
 * constructor and both interfaces in the class's own implements clause, as CodecHolder and
 * addCompressionCodec require.  Not compiled in the build image (no JDK); see INTEGRATION.md.
 */
package org.gridfour.hip;

import java.io.IOException;
import java.io.PrintStream;
import org.gridfour.compress.CodecFloat;
import org.gridfour.compress.ICompressionDecoder;
import org.gridfour.compress.ICompressionEncoder;

public class CodecFloatHip implements ICompressionEncoder, ICompressionDecoder {

  private final long handle = HipCodecNative.create(Integer.getInteger("gridfour.hip.device", 0));
  private final int level = Integer.getInteger("gridfour.hip.floatLevel", 9);
  /** analysis statistics are host-side bookkeeping: delegate to the stock implementation */
  private final CodecFloat statsDelegate = new CodecFloat();

  public CodecFloatHip() {
  }

  @Override
  public byte[] encode(int codecIndex, int nRows, int nCols, int[] values) {
    throw new IllegalArgumentException(
      "Attempt to enccode an integral format not supported by this CODEC");   // CodecFloat.java:121-125
  }

  @Override
  public int[] decode(int nRows, int nColumns, byte[] packing) throws IOException {
    throw new IOException(
      "Attempt to decode an integral format not supported by this CODEC");    // CodecFloat.java:116-119
  }

  @Override
  public byte[] encodeFloats(int codecIndex, int nRows, int nCols, float[] values) {
    return HipCodecNative.encodeFloats(handle, codecIndex, nRows, nCols, values, level);
  }

  @Override
  public float[] decodeFloats(int nRows, int nColumns, byte[] packing) throws IOException {
    return HipCodecNative.decodeFloats(handle, nRows, nColumns, packing);
  }

  @Override
  public boolean implementsFloatingPointEncoding() {
    return true;
  }

  @Override
  public boolean implementsIntegerEncoding() {
    return false;
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
