// jni.h -- TEST INFRASTRUCTURE ONLY: a stand-in JVM for gridfour_amd/java/gvrs_hip_jni.cpp.
//
// The build image holds no JDK, so the JNI shim of this repository (our own code, not the reference's) had never met a compiler.
// This header declares the part of the Java Native Interface the shim uses -- the primitive types, the array class hierarchy of
// the C++ binding, JNIEnv's member functions FindClass / ThrowNew / GetArrayLength / New<T>Array / Get<T>ArrayRegion /
// Set<T>ArrayRegion with the signatures of the JNI specification (chapter 4) -- and implements them over plain heap objects, so
// that tests/csrc/jni_shim_test.cpp can CALL the shim's Java_* functions the way a JVM would and check what comes back.
// It proves the shim compiles, links against libgvrs_hip.so and moves arrays / raises exceptions as intended; it proves nothing
// about a real JVM (no local-reference tables, no GC, no class loading) and nothing about parity with the Java codec.
#ifndef GVRS_TEST_JNI_MOCK_H
#define GVRS_TEST_JNI_MOCK_H

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_FALSE 0
#define JNI_TRUE 1

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef uint8_t jboolean;
typedef uint16_t jchar;
typedef int16_t jshort;
typedef float jfloat;
typedef double jdouble;
typedef jint jsize;

// the C++ binding's class hierarchy (jni.h of a JDK: class _jobject {}; class _jclass : public _jobject {}; ...)
class _jobject {
public:
    virtual ~_jobject() {}
};
class _jclass : public _jobject {
public:
    std::string name;
};
class _jthrowable : public _jobject {};
class _jstring : public _jobject {};
class _jarray : public _jobject {
public:
    size_t elemBytes = 0;
    jsize length = 0;
    std::vector<uint8_t> bytes;
};
class _jbyteArray : public _jarray {};
class _jshortArray : public _jarray {};
class _jintArray : public _jarray {};
class _jlongArray : public _jarray {};
class _jfloatArray : public _jarray {};
typedef _jobject *jobject;
typedef _jclass *jclass;
typedef _jthrowable *jthrowable;
typedef _jstring *jstring;
typedef _jarray *jarray;
typedef _jbyteArray *jbyteArray;
typedef _jshortArray *jshortArray;
typedef _jintArray *jintArray;
typedef _jlongArray *jlongArray;
typedef _jfloatArray *jfloatArray;

struct JNIEnv_ {
    // what the stand-in JVM records (a real JVM keeps the pending exception per thread: one JNIEnv_ per test thread here)
    std::string pendingClass, pendingMessage;
    std::vector<_jobject *> owned;
    long outOfBounds = 0;                                    // region calls that a JVM would answer with ArrayIndexOutOfBoundsException

    ~JNIEnv_()
    {
        for (_jobject *o : owned) delete o;
    }
    bool ExceptionCheck() const { return !pendingClass.empty(); }
    void ExceptionClear()
    {
        pendingClass.clear();
        pendingMessage.clear();
    }

    jclass FindClass(const char *name)
    {
        _jclass *c = new _jclass;
        c->name = name;
        owned.push_back(c);
        return c;
    }
    jint ThrowNew(jclass c, const char *msg)
    {
        pendingClass = c ? c->name : "?";
        pendingMessage = msg ? msg : "";
        return 0;
    }
    jsize GetArrayLength(jarray a) { return a->length; }

    template <class A> A *newArray(jsize n, size_t elemBytes)
    {
        if (n < 0) return nullptr;
        A *a = new A;
        a->elemBytes = elemBytes;
        a->length = n;
        a->bytes.assign((size_t)n * elemBytes, 0);
        owned.push_back(a);
        return a;
    }
    jbyteArray NewByteArray(jsize n) { return newArray<_jbyteArray>(n, 1); }
    jshortArray NewShortArray(jsize n) { return newArray<_jshortArray>(n, 2); }
    jintArray NewIntArray(jsize n) { return newArray<_jintArray>(n, 4); }
    jlongArray NewLongArray(jsize n) { return newArray<_jlongArray>(n, 8); }
    jfloatArray NewFloatArray(jsize n) { return newArray<_jfloatArray>(n, 4); }

    bool region(_jarray *a, jsize start, jsize len, size_t elemBytes)
    {
        if (!a || a->elemBytes != elemBytes || start < 0 || len < 0 || (int64_t)start + len > a->length) {
            outOfBounds++;
            pendingClass = "java/lang/ArrayIndexOutOfBoundsException";
            return false;
        }
        return true;
    }
    void get(_jarray *a, jsize start, jsize len, void *buf, size_t eb)
    {
        if (region(a, start, len, eb)) memcpy(buf, a->bytes.data() + (size_t)start * eb, (size_t)len * eb);
    }
    void set(_jarray *a, jsize start, jsize len, const void *buf, size_t eb)
    {
        if (region(a, start, len, eb)) memcpy(a->bytes.data() + (size_t)start * eb, buf, (size_t)len * eb);
    }
    void GetByteArrayRegion(jbyteArray a, jsize s, jsize l, jbyte *buf) { get(a, s, l, buf, 1); }
    void GetShortArrayRegion(jshortArray a, jsize s, jsize l, jshort *buf) { get(a, s, l, buf, 2); }
    void GetIntArrayRegion(jintArray a, jsize s, jsize l, jint *buf) { get(a, s, l, buf, 4); }
    void GetLongArrayRegion(jlongArray a, jsize s, jsize l, jlong *buf) { get(a, s, l, buf, 8); }
    void GetFloatArrayRegion(jfloatArray a, jsize s, jsize l, jfloat *buf) { get(a, s, l, buf, 4); }
    void SetByteArrayRegion(jbyteArray a, jsize s, jsize l, const jbyte *buf) { set(a, s, l, buf, 1); }
    void SetShortArrayRegion(jshortArray a, jsize s, jsize l, const jshort *buf) { set(a, s, l, buf, 2); }
    void SetIntArrayRegion(jintArray a, jsize s, jsize l, const jint *buf) { set(a, s, l, buf, 4); }
    void SetLongArrayRegion(jlongArray a, jsize s, jsize l, const jlong *buf) { set(a, s, l, buf, 8); }
    void SetFloatArrayRegion(jfloatArray a, jsize s, jsize l, const jfloat *buf) { set(a, s, l, buf, 4); }
};
typedef JNIEnv_ JNIEnv;

#endif
