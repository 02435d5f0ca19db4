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

  private HipCodecNative() {
  }
}
