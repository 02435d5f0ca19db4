/*
 * Adapter that registers the MI355X implementation of the LSOP12 codec (LsEncoder12 + LsDecoder12,
 * lsop/LsCodecUtility.java:53-64) under Gridfour's plug-in interface.
 *
 *   spec.addCompressionCodec("LSOP12", org.gridfour.hip.LsCodecHip.class);
 *
 * The packings are byte-identical to LsEncoder12's, so files stay readable by stock Gridfour
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
import org.gridfour.lsop.LsDecoder12;

public class LsCodecHip implements ICompressionEncoder, ICompressionDecoder {

  /** LsEncoder12's two switches (setDeflateEnabled, default on; setValueChecksumEnabled, default off); system properties preset them */
  private boolean deflateEnabled = Boolean.parseBoolean(System.getProperty("gridfour.hip.lsop.deflate", "true"));
  private boolean valueChecksumEnabled = Boolean.parseBoolean(System.getProperty("gridfour.hip.lsop.checksum", "false"));
  private final long handle = HipCodecNative.create(Integer.getInteger("gridfour.hip.device", 0));
  /** analysis statistics are host-side bookkeeping: delegate to the stock implementation */
  private final LsDecoder12 statsDelegate = new LsDecoder12();

  public LsCodecHip() {
  }

  /** as LsEncoder12.setDeflateEnabled (lsop/LsEncoder12.java:92-94) */
  public void setDeflateEnabled(boolean deflateEnabled) {
    this.deflateEnabled = deflateEnabled;
  }

  /** as LsEncoder12.setValueChecksumEnabled (lsop/LsEncoder12.java:117-119) */
  public void setValueChecksumEnabled(boolean valueChecksumEnabled) {
    this.valueChecksumEnabled = valueChecksumEnabled;
  }

  /** the native kind: 2 / 3 = LSOP12 without / with the Deflate alternative, 5 / 6 = the same with the value checksum */
  private int kind() {
    return (deflateEnabled ? 3 : 2) + (valueChecksumEnabled ? 3 : 0);
  }

  @Override
  public byte[] encode(int codecIndex, int nRows, int nCols, int[] values) {
    return HipCodecNative.encode(handle, kind(), codecIndex, nRows, nCols, values);
  }

  @Override
  public int[] decode(int nRows, int nColumns, byte[] packing) throws IOException {
    return HipCodecNative.decode(handle, kind(), nRows, nColumns, packing);
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
