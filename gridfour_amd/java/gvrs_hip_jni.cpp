// gvrs_hip_jni.cpp -- thin JNI shim between org.gridfour.hip.CodecHuffmanHip (Java adapter in this
// directory) and the C ABI of libgvrs_hip.so (include/gvrs_hip_codec.h).
//
// Not compiled in the build image (no JDK, no jni.h).  Build on a host with a JDK:
//   g++ -O2 -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I<repo>/include
//       gvrs_hip_jni.cpp -L<repo>/gridfour_amd/lib -lgvrs_hip -o libgvrs_hip_jni.so
// One gf_context per adapter instance; CodecHolder creates one instance per codec
// (gvrs/CodecHolder.java:208-234) and the decoder may be called from two threads
// (gvrs/TileDecompressionAssistant.java:68-73), hence the mutex around context use.
// Java arrays are copied to and from native buffers with Get/Set<Type>ArrayRegion OUTSIDE the lock: no JNI critical region
// is ever open while this code waits for the mutex, copies over PCIe, launches kernels or synchronises a stream (the JNI
// specification forbids blocking inside GetPrimitiveArrayCritical regions).
#include <jni.h>

#include <mutex>
#include <vector>

#include "gvrs_hip_codec.h"

namespace {
struct Handle {
    gf_context *ctx = nullptr;
    std::mutex lock;
};
void throwIo(JNIEnv *env, const char *msg)
{
    jclass c = env->FindClass("java/io/IOException");
    if (c) env->ThrowNew(c, msg);
}
// the handle of a native method: null (a create that failed, a closed adapter) raises IllegalStateException
Handle *handleOf(JNIEnv *env, jlong handle)
{
    Handle *h = (Handle *)(intptr_t)handle;
    if (!h || !h->ctx) {
        jclass c = env->FindClass("java/lang/IllegalStateException");
        if (c) env->ThrowNew(c, "the native HIP codec is not open");
        return nullptr;
    }
    return h;
}
std::vector<int32_t> intsOf(JNIEnv *env, jintArray a)
{
    const jsize n = a ? env->GetArrayLength(a) : 0;
    std::vector<int32_t> v((size_t)n + 1);
    if (n) env->GetIntArrayRegion(a, 0, n, (jint *)v.data());
    return v;
}
std::vector<uint8_t> bytesOf(JNIEnv *env, jbyteArray a)
{
    const jsize n = a ? env->GetArrayLength(a) : 0;
    std::vector<uint8_t> v((size_t)n + 16);          // the decoders read whole words
    if (n) env->GetByteArrayRegion(a, 0, n, (jbyte *)v.data());
    return v;
}
// cells of a tile batch: int[] (GF_ELEM_INT) or short[] (GF_ELEM_SHORT)
std::vector<uint8_t> cellsOf(JNIEnv *env, jobject cells, int elemType)
{
    const jsize n = env->GetArrayLength((jarray)cells);
    const size_t w = elemType == GF_ELEM_SHORT ? 2 : 4;
    std::vector<uint8_t> v((size_t)n * w + 16);
    if (n) {
        if (elemType == GF_ELEM_SHORT) env->GetShortArrayRegion((jshortArray)cells, 0, n, (jshort *)v.data());
        else env->GetIntArrayRegion((jintArray)cells, 0, n, (jint *)v.data());
    }
    return v;
}
}  // namespace

extern "C" {

JNIEXPORT jlong JNICALL Java_org_gridfour_hip_CodecHuffmanHip_createNative(JNIEnv *env, jclass, jint device)
{
    Handle *h = new Handle();
    if (gf_context_create(device, &h->ctx) != GF_OK) {
        jclass c = env->FindClass("java/lang/IllegalStateException");
        if (c) env->ThrowNew(c, gf_last_error());
        delete h;
        return 0;
    }
    return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL Java_org_gridfour_hip_CodecHuffmanHip_destroyNative(JNIEnv *, jclass, jlong handle)
{
    Handle *h = (Handle *)(intptr_t)handle;
    if (!h) return;
    gf_context_destroy(h->ctx);
    delete h;
}

// byte[] encode(int codecIndex, int nRows, int nCols, int[] values)  -> null when declined
JNIEXPORT jbyteArray JNICALL Java_org_gridfour_hip_CodecHuffmanHip_encodeNative(JNIEnv *env, jclass, jlong handle, jint codecIndex,
                                                                               jint nRows, jint nCols, jintArray values)
{
    Handle *h = handleOf(env, handle);
    if (!h) return nullptr;
    const size_t cap = gf_huffman_max_packing(nRows, nCols);
    jbyte *out = new jbyte[cap];
    size_t n = 0;
    gf_status s;
    const std::vector<int32_t> v = intsOf(env, values);
    {
        std::lock_guard<std::mutex> g(h->lock);
        s = gf_huffman_encode_i32(h->ctx, codecIndex, nRows, nCols, v.data(), (uint8_t *)out, cap, &n);
    }
    jbyteArray result = nullptr;
    if (s == GF_OK) {
        result = env->NewByteArray((jsize)n);
        if (result) env->SetByteArrayRegion(result, 0, (jsize)n, out);
    } else if (s == GF_ERR_BOUNDS) {
        jclass c = env->FindClass("java/lang/ArrayIndexOutOfBoundsException");
        if (c) env->ThrowNew(c, "tile has fewer than 2 columns");
    } else if (s != GF_DECLINED) {
        jclass c = env->FindClass("java/lang/IllegalStateException");
        if (c) env->ThrowNew(c, gf_status_string(s));
    }
    delete[] out;
    return result;   // null == Java null (CodecHuffman.java:80-82)
}

// int[] decode(int nRows, int nColumns, byte[] packing) throws IOException
JNIEXPORT jintArray JNICALL Java_org_gridfour_hip_CodecHuffmanHip_decodeNative(JNIEnv *env, jclass, jlong handle, jint nRows,
                                                                              jint nCols, jbyteArray packing)
{
    Handle *h = handleOf(env, handle);
    if (!h) return nullptr;
    const jsize len = env->GetArrayLength(packing);
    const std::vector<uint8_t> p = bytesOf(env, packing);
    std::vector<int32_t> o((size_t)nRows * (size_t)nCols + 1);
    gf_status s;
    {
        std::lock_guard<std::mutex> g(h->lock);
        s = gf_huffman_decode_i32(h->ctx, nRows, nCols, p.data(), (size_t)len, o.data());
    }
    if (s != GF_OK) {
        throwIo(env, gf_status_string(s));
        return nullptr;
    }
    jintArray result = env->NewIntArray(nRows * nCols);
    if (result) env->SetIntArrayRegion(result, 0, nRows * nCols, (const jint *)o.data());
    return result;
}


// ---- org.gridfour.hip.HipCodecNative: the same two calls for every integer codec of the library ----
// kind 0 = CodecHuffman, 1 = CodecCanonHuffman, 2 = LSOP12 (Deflate alternative disabled), 3 = LSOP12 (reference default),
// 4 = CodecDeflate, 5 / 6 = LSOP12 as 2 / 3 with the value checksum in the header (LsEncoder12.setValueChecksumEnabled)
JNIEXPORT jlong JNICALL Java_org_gridfour_hip_HipCodecNative_create(JNIEnv *env, jclass cls, jint device)
{
    return Java_org_gridfour_hip_CodecHuffmanHip_createNative(env, cls, device);
}

JNIEXPORT void JNICALL Java_org_gridfour_hip_HipCodecNative_destroy(JNIEnv *env, jclass cls, jlong handle)
{
    Java_org_gridfour_hip_CodecHuffmanHip_destroyNative(env, cls, handle);
}

JNIEXPORT jbyteArray JNICALL Java_org_gridfour_hip_HipCodecNative_encode(JNIEnv *env, jclass, jlong handle, jint kind,
                                                                         jint codecIndex, jint nRows, jint nCols, jintArray values)
{
    Handle *h = handleOf(env, handle);
    if (!h) return nullptr;
    const size_t cap = kind == 0 ? gf_huffman_max_packing(nRows, nCols)
                     : kind == 1 ? gf_canon_max_packing(nRows, nCols)
                     : kind == 4 ? gf_m32_max_stream(nRows, nCols) + 256 : gf_lsop12_max_packing(nRows, nCols) + 64;
    jbyte *out = new jbyte[cap];
    size_t n = 0;
    gf_status s;
    const std::vector<int32_t> vals = intsOf(env, values);
    const int32_t *v = vals.data();
    {
        std::lock_guard<std::mutex> g(h->lock);
        if (kind == 0) s = gf_huffman_encode_i32(h->ctx, codecIndex, nRows, nCols, v, (uint8_t *)out, cap, &n);
        else if (kind == 1) s = gf_canon_encode_i32(h->ctx, codecIndex, nRows, nCols, v, (uint8_t *)out, cap, &n);
        else if (kind == 4) s = gf_deflate_encode_i32(h->ctx, codecIndex, nRows, nCols, v, (uint8_t *)out, cap, &n);
        else s = gf_lsop12_encode_i32(h->ctx, codecIndex, nRows, nCols, v,
                                      ((kind == 3 || kind == 6) ? GF_LSOP_DEFLATE : 0) | (kind >= 5 ? GF_LSOP_VALUE_CHECKSUM : 0),
                                      (uint8_t *)out, cap, &n);
    }
    jbyteArray result = nullptr;
    if (s == GF_OK) {
        result = env->NewByteArray((jsize)n);
        if (result) env->SetByteArrayRegion(result, 0, (jsize)n, out);
    } else if (s == GF_ERR_BOUNDS) {
        jclass c = env->FindClass("java/lang/ArrayIndexOutOfBoundsException");
        if (c) env->ThrowNew(c, "tile has fewer than 2 columns");
    } else if (s == GF_ERR_ARG && kind == 1) {
        jclass c = env->FindClass("java/lang/IllegalArgumentException");      // CanonicalHuffman.java:183-185
        if (c) env->ThrowNew(c, "Empty or null data input data");
    } else if (s != GF_DECLINED) {
        jclass c = env->FindClass("java/lang/IllegalStateException");
        if (c) env->ThrowNew(c, gf_status_string(s));
    }
    delete[] out;
    return result;
}

JNIEXPORT jintArray JNICALL Java_org_gridfour_hip_HipCodecNative_decode(JNIEnv *env, jclass, jlong handle, jint kind, jint nRows,
                                                                        jint nCols, jbyteArray packing)
{
    Handle *h = handleOf(env, handle);
    if (!h) return nullptr;
    const jsize len = env->GetArrayLength(packing);
    const std::vector<uint8_t> pk = bytesOf(env, packing);
    const uint8_t *p = pk.data();
    std::vector<int32_t> out((size_t)nRows * (size_t)nCols + 1);
    int32_t *o = out.data();
    gf_status s;
    {
        std::lock_guard<std::mutex> g(h->lock);
        if (kind == 0) s = gf_huffman_decode_i32(h->ctx, nRows, nCols, p, (size_t)len, o);
        else if (kind == 1) s = gf_canon_decode_i32(h->ctx, nRows, nCols, p, (size_t)len, o);
        else if (kind == 4) s = gf_deflate_decode_i32(h->ctx, nRows, nCols, p, (size_t)len, o);
        else s = gf_lsop12_decode_i32(h->ctx, nRows, nCols, p, (size_t)len, o);
    }
    if (s == GF_DECLINED) return nullptr;          // CodecDeflate.decode: the inflater gave nothing -> null (CodecDeflate.java:143-154)
    if (s != GF_OK) {
        throwIo(env, gf_status_string(s));
        return nullptr;
    }
    jintArray result = env->NewIntArray(nRows * nCols);
    if (result) env->SetIntArrayRegion(result, 0, nRows * nCols, (const jint *)o);
    return result;
}

// byte[] encodeFloats(long handle, int codecIndex, int nRows, int nCols, float[] values, int level): ICompressionEncoder.encodeFloats
// as CodecFloat implements it (compress/CodecFloat.java:328-369, ICompressionEncoder.java:76-91)
JNIEXPORT jbyteArray JNICALL Java_org_gridfour_hip_HipCodecNative_encodeFloats(JNIEnv *env, jclass, jlong handle, jint codecIndex,
                                                                               jint nRows, jint nCols, jfloatArray values, jint level)
{
    Handle *h = handleOf(env, handle);
    if (!h) return nullptr;
    const size_t cells = (size_t)nRows * (size_t)nCols;
    if (!values || (size_t)env->GetArrayLength(values) < cells) {
        jclass c = env->FindClass("java/lang/ArrayIndexOutOfBoundsException");   // values[k] beyond the array in the reference's loops
        if (c) env->ThrowNew(c, "fewer values than nRows * nCols");
        return nullptr;
    }
    std::vector<float> v(cells + 1);
    if (cells) env->GetFloatArrayRegion(values, 0, (jsize)cells, (jfloat *)v.data());
    // 2 header bytes + five planes, each a length and a zlib stream (stored blocks at worst: a few bytes per 16 KB more)
    const size_t cap = 5 * cells + 4096;
    std::vector<uint8_t> out(cap);
    size_t n = 0;
    gf_status s;
    {
        std::lock_guard<std::mutex> g(h->lock);
        s = gf_float_encode_f32(h->ctx, codecIndex, nRows, nCols, v.data(), level, out.data(), cap, &n);
    }
    if (s != GF_OK) {
        jclass c = env->FindClass("java/lang/IllegalStateException");
        if (c) env->ThrowNew(c, gf_status_string(s));
        return nullptr;
    }
    jbyteArray result = env->NewByteArray((jsize)n);
    if (result) env->SetByteArrayRegion(result, 0, (jsize)n, (const jbyte *)out.data());
    return result;
}

// float[] decodeFloats(long handle, int nRows, int nColumns, byte[] packing): ICompressionDecoder.decodeFloats as CodecFloat
// implements it (compress/CodecFloat.java:371-458); a stream the inflater rejects is CodecFloat's IOException
JNIEXPORT jfloatArray JNICALL Java_org_gridfour_hip_HipCodecNative_decodeFloats(JNIEnv *env, jclass, jlong handle, jint nRows,
                                                                                jint nCols, jbyteArray packing)
{
    Handle *h = handleOf(env, handle);
    if (!h) return nullptr;
    const jsize len = env->GetArrayLength(packing);
    const std::vector<uint8_t> pk = bytesOf(env, packing);
    const size_t cells = (size_t)nRows * (size_t)nCols;
    std::vector<float> out(cells + 1);
    gf_status s;
    {
        std::lock_guard<std::mutex> g(h->lock);
        s = gf_float_decode_f32(h->ctx, nRows, nCols, pk.data(), (size_t)len, out.data());
    }
    if (s != GF_OK) {
        throwIo(env, gf_status_string(s));
        return nullptr;
    }
    jfloatArray result = env->NewFloatArray((jsize)cells);
    if (result) env->SetFloatArrayRegion(result, 0, (jsize)cells, (const jfloat *)out.data());
    return result;
}

// byte[] tileRecords(long handle, int[] codecKinds, int elemType, int fillValue, int nRows, int nCols, int[] tileIndices,
//                    Object cells, boolean checksums, long[] recordOffsets)
JNIEXPORT jbyteArray JNICALL Java_org_gridfour_hip_HipCodecNative_tileRecords(JNIEnv *env, jclass, jlong handle, jintArray codecKinds,
                                                                              jint elemType, jint fillValue, jint nRows, jint nCols,
                                                                              jintArray tileIndices, jobject cells, jboolean checksums,
                                                                              jlongArray recordOffsets)
{
    Handle *h = handleOf(env, handle);
    if (!h) return nullptr;
    const jsize nTiles = env->GetArrayLength(tileIndices), nCodecs = env->GetArrayLength(codecKinds);
    if (env->GetArrayLength(recordOffsets) < nTiles + 1) return nullptr;
    const size_t cap = (size_t)nTiles * gf_tile_record_max_bytes(elemType, nRows, nCols);
    std::vector<uint8_t> blob(cap ? cap : 8);
    std::vector<uint64_t> offsets((size_t)nTiles + 1);
    std::vector<int> kinds((size_t)nCodecs + 1);
    env->GetIntArrayRegion(codecKinds, 0, nCodecs, (jint *)kinds.data());
    std::vector<int32_t> idx((size_t)nTiles + 1);
    env->GetIntArrayRegion(tileIndices, 0, nTiles, (jint *)idx.data());
    gf_status s;
    const std::vector<uint8_t> v = cellsOf(env, cells, elemType);
    {
        std::lock_guard<std::mutex> g(h->lock);
        s = gf_tile_record_encode_batch(h->ctx, kinds.data(), nCodecs, elemType, fillValue, nRows, nCols, (size_t)nTiles, idx.data(),
                                        v.data(), checksums ? 1 : 0, blob.data(), cap, offsets.data(), nullptr);
    }
    if (s != GF_OK) {
        jclass c = env->FindClass("java/io/IOException");
        if (c) env->ThrowNew(c, gf_last_error());
        return nullptr;
    }
    env->SetLongArrayRegion(recordOffsets, 0, nTiles + 1, (const jlong *)offsets.data());
    jbyteArray result = env->NewByteArray((jsize)offsets[nTiles]);
    if (result) env->SetByteArrayRegion(result, 0, (jsize)offsets[nTiles], (const jbyte *)blob.data());
    return result;
}

// void tilesFromRecords(long handle, int[] codecKinds, int elemType, int nRows, int nCols, byte[] records, long[] recordOffsets,
//                       boolean verifyChecksums, int[] tileIndices, Object cells, int[] status)
JNIEXPORT void JNICALL Java_org_gridfour_hip_HipCodecNative_tilesFromRecords(JNIEnv *env, jclass, jlong handle, jintArray codecKinds,
                                                                             jint elemType, jint nRows, jint nCols, jbyteArray records,
                                                                             jlongArray recordOffsets, jboolean verifyChecksums,
                                                                             jintArray tileIndices, jobject cells, jintArray status)
{
    Handle *h = handleOf(env, handle);
    if (!h) return;
    const jsize nTiles = env->GetArrayLength(tileIndices), nCodecs = env->GetArrayLength(codecKinds);
    std::vector<int> kinds((size_t)nCodecs + 1);
    env->GetIntArrayRegion(codecKinds, 0, nCodecs, (jint *)kinds.data());
    if (env->GetArrayLength(recordOffsets) < nTiles + 1 || env->GetArrayLength(status) < nTiles) {
        jclass c = env->FindClass("java/lang/IllegalArgumentException");
        if (c) env->ThrowNew(c, "recordOffsets needs nTiles + 1 entries, status nTiles");
        return;
    }
    std::vector<uint64_t> offsets((size_t)nTiles + 1);
    env->GetLongArrayRegion(recordOffsets, 0, nTiles + 1, (jlong *)offsets.data());
    std::vector<int32_t> idx((size_t)nTiles + 1), st((size_t)nTiles + 1);
    const jsize len = env->GetArrayLength(records);
    std::vector<uint8_t> blob((size_t)len + 16);
    env->GetByteArrayRegion(records, 0, len, (jbyte *)blob.data());
    // the library checks that the offsets are monotone and inside the record bytes (GF_ERR_ARG otherwise)
    if (offsets[(size_t)nTiles] > (uint64_t)len) {
        jclass c = env->FindClass("java/lang/IllegalArgumentException");
        if (c) env->ThrowNew(c, "recordOffsets run past the record bytes");
        return;
    }
    gf_status s;
    const jsize nCells = env->GetArrayLength((jarray)cells);
    std::vector<uint8_t> v((size_t)nCells * (elemType == GF_ELEM_SHORT ? 2 : 4) + 16);
    {
        std::lock_guard<std::mutex> g(h->lock);
        s = gf_tile_record_decode_batch(h->ctx, kinds.data(), nCodecs, elemType, nRows, nCols, (size_t)nTiles, blob.data(), offsets.data(),
                                        verifyChecksums ? 1 : 0, idx.data(), v.data(), st.data());
    }
    if (s != GF_OK) {
        jclass c = env->FindClass("java/io/IOException");
        if (c) env->ThrowNew(c, gf_last_error());
        return;
    }
    if (elemType == GF_ELEM_SHORT) env->SetShortArrayRegion((jshortArray)cells, 0, nCells, (const jshort *)v.data());
    else env->SetIntArrayRegion((jintArray)cells, 0, nCells, (const jint *)v.data());
    env->SetIntArrayRegion(tileIndices, 0, nTiles, (const jint *)idx.data());
    env->SetIntArrayRegion(status, 0, nTiles, (const jint *)st.data());
}

// ---- read-ahead: org.gridfour.hip.HipReadAhead over gf_readahead_* ----
JNIEXPORT jlong JNICALL Java_org_gridfour_hip_HipCodecNative_readaheadCreate(JNIEnv *env, jclass, jint device, jintArray codecKinds,
                                                                           jint nRows, jint nCols, jint maxBatch)
{
    const std::vector<int32_t> kinds = intsOf(env, codecKinds);
    const jsize nCodecs = codecKinds ? env->GetArrayLength(codecKinds) : 0;
    gf_readahead *ra = nullptr;
    if (gf_readahead_create(device, (const int *)kinds.data(), nCodecs, nRows, nCols, maxBatch > 0 ? (size_t)maxBatch : 1, &ra) != GF_OK) {
        jclass c = env->FindClass("java/lang/IllegalStateException");
        if (c) env->ThrowNew(c, gf_last_error());
        return 0;
    }
    return (jlong)(intptr_t)ra;
}

JNIEXPORT void JNICALL Java_org_gridfour_hip_HipCodecNative_readaheadDestroy(JNIEnv *, jclass, jlong ra)
{
    gf_readahead_destroy((gf_readahead *)(intptr_t)ra);            // (joins the worker; null is a no-op)
}

JNIEXPORT void JNICALL Java_org_gridfour_hip_HipCodecNative_readaheadSubmit(JNIEnv *env, jclass, jlong ra, jint tileIndex,
                                                                          jbyteArray packing)
{
    if (!ra || !packing) return;
    const jsize len = env->GetArrayLength(packing);
    const std::vector<uint8_t> p = bytesOf(env, packing);
    gf_readahead_submit((gf_readahead *)(intptr_t)ra, tileIndex, p.data(), (size_t)len);    // the library copies the bytes
}

JNIEXPORT jint JNICALL Java_org_gridfour_hip_HipCodecNative_readaheadPending(JNIEnv *, jclass, jlong ra)
{
    return ra ? gf_readahead_pending((gf_readahead *)(intptr_t)ra) : 0;
}

JNIEXPORT jint JNICALL Java_org_gridfour_hip_HipCodecNative_readaheadTake(JNIEnv *env, jclass, jlong ra, jint waitIndex,
                                                                        jintArray indices, jintArray cells, jintArray status)
{
    if (!ra) return 0;
    const jsize maxTiles = env->GetArrayLength(indices), nCells = env->GetArrayLength(cells);
    if (maxTiles == 0) return 0;
    // gf_readahead_take writes `per` ints for every tile it hands over: the arrays must have room for maxTiles of them (a direct
    // call of this static native with a shorter array would otherwise run over the native buffer)
    const size_t per = gf_readahead_cells((gf_readahead *)(intptr_t)ra);
    if ((size_t)nCells < (size_t)maxTiles * per || env->GetArrayLength(status) < maxTiles) {
        jclass c = env->FindClass("java/lang/IllegalArgumentException");
        if (c) env->ThrowNew(c, "readaheadTake: cells needs indices.length * nRows * nCols ints, status indices.length");
        return 0;
    }
    // (the wait happens here, in native code, with no Java array pinned)
    std::vector<int32_t> idx((size_t)maxTiles), st((size_t)maxTiles), v((size_t)maxTiles * per + 1);
    size_t n = 0;
    const gf_status s = gf_readahead_take((gf_readahead *)(intptr_t)ra, waitIndex, (size_t)maxTiles, idx.data(), v.data(), st.data(), &n);
    if (s != GF_OK) {
        jclass c = env->FindClass("java/lang/IllegalStateException");
        if (c) env->ThrowNew(c, gf_status_string(s));
        return 0;
    }
    if (n) {
        env->SetIntArrayRegion(indices, 0, (jsize)n, (const jint *)idx.data());
        env->SetIntArrayRegion(status, 0, (jsize)n, (const jint *)st.data());
        env->SetIntArrayRegion(cells, 0, (jsize)(n * per), (const jint *)v.data());
    }
    return (jint)n;
}

// ---- several GPUs from one JVM: org.gridfour.hip.HipMultiGpu over gf_multi_* ----
JNIEXPORT jlong JNICALL Java_org_gridfour_hip_HipCodecNative_multiCreate(JNIEnv *env, jclass, jintArray devices)
{
    const std::vector<int32_t> dev = intsOf(env, devices);
    gf_multi *m = nullptr;
    if (gf_multi_create((const int *)dev.data(), devices ? env->GetArrayLength(devices) : 0, &m) != GF_OK) {
        jclass c = env->FindClass("java/lang/IllegalStateException");
        if (c) env->ThrowNew(c, gf_last_error());
        return 0;
    }
    return (jlong)(intptr_t)m;
}

JNIEXPORT void JNICALL Java_org_gridfour_hip_HipCodecNative_multiDestroy(JNIEnv *, jclass, jlong multi)
{
    gf_multi_destroy((gf_multi *)(intptr_t)multi);
}

JNIEXPORT jbyteArray JNICALL Java_org_gridfour_hip_HipCodecNative_multiHuffmanEncode(JNIEnv *env, jclass, jlong multi, jint codecIndex,
                                                                                   jint nRows, jint nCols, jintArray cells,
                                                                                   jlongArray offsets, jbyteArray predictors,
                                                                                   jintArray status)
{
    if (!multi) return nullptr;
    const size_t per = (size_t)nRows * (size_t)nCols;
    const jsize nCells = env->GetArrayLength(cells);
    const size_t nTiles = per ? (size_t)nCells / per : 0;
    if (env->GetArrayLength(offsets) < (jsize)nTiles + 1) return nullptr;
    const std::vector<int32_t> v = intsOf(env, cells);
    std::vector<uint64_t> off(nTiles + 1);
    std::vector<uint8_t> pred(nTiles + 1);
    std::vector<int32_t> st(nTiles + 1);
    size_t cap = nTiles * per + 4096;                                // a byte per cell holds terrain packings; grown on demand
    std::vector<uint8_t> blob;
    gf_status s;
    for (int attempt = 0; attempt < 2; attempt++) {                  // one regrow at most: the call reports the size it needs
        blob.resize(cap);
        s = gf_huffman_encode_batch_i32_multi((gf_multi *)(intptr_t)multi, codecIndex, nRows, nCols, nTiles, v.data(), blob.data(), cap,
                                              off.data(), pred.data(), st.data());
        if (s != GF_ERR_CAPACITY || off[nTiles] == 0) break;
        cap = (size_t)off[nTiles] + 64;
    }
    if (s != GF_OK) {
        throwIo(env, gf_status_string(s));
        return nullptr;
    }
    env->SetLongArrayRegion(offsets, 0, (jsize)nTiles + 1, (const jlong *)off.data());
    if (predictors) env->SetByteArrayRegion(predictors, 0, (jsize)nTiles, (const jbyte *)pred.data());
    if (status) env->SetIntArrayRegion(status, 0, (jsize)nTiles, (const jint *)st.data());
    jbyteArray result = env->NewByteArray((jsize)off[nTiles]);
    if (result) env->SetByteArrayRegion(result, 0, (jsize)off[nTiles], (const jbyte *)blob.data());
    return result;
}

JNIEXPORT void JNICALL Java_org_gridfour_hip_HipCodecNative_multiHuffmanDecode(JNIEnv *env, jclass, jlong multi, jint nRows, jint nCols,
                                                                             jbyteArray blob, jlongArray offsets, jintArray cells,
                                                                             jintArray status)
{
    if (!multi) return;
    const size_t per = (size_t)nRows * (size_t)nCols;
    const jsize nOff = env->GetArrayLength(offsets);
    if (nOff < 1) return;
    const size_t nTiles = (size_t)nOff - 1;
    if ((size_t)env->GetArrayLength(cells) < nTiles * per) {
        jclass c = env->FindClass("java/lang/IllegalArgumentException");
        if (c) env->ThrowNew(c, "cells must hold nTiles x nRows x nCols values");
        return;
    }
    std::vector<uint64_t> off((size_t)nOff);
    env->GetLongArrayRegion(offsets, 0, nOff, (jlong *)off.data());
    const std::vector<uint8_t> b = bytesOf(env, blob);
    if (off[nTiles] > (uint64_t)env->GetArrayLength(blob)) {
        jclass c = env->FindClass("java/lang/IllegalArgumentException");
        if (c) env->ThrowNew(c, "offsets run past the packings");
        return;
    }
    std::vector<int32_t> v(nTiles * per + 1), st(nTiles + 1);
    const gf_status s = gf_huffman_decode_batch_i32_multi((gf_multi *)(intptr_t)multi, nRows, nCols, nTiles, b.data(), off.data(),
                                                          v.data(), st.data());
    if (s != GF_OK) {
        throwIo(env, gf_status_string(s));
        return;
    }
    env->SetIntArrayRegion(cells, 0, (jsize)(nTiles * per), (const jint *)v.data());
    if (status) env->SetIntArrayRegion(status, 0, (jsize)nTiles, (const jint *)st.data());
}

}  // extern "C"
