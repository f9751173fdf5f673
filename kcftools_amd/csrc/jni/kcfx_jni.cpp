// kcfx_jni.cpp -- thin JNI shim over the C ABI of include/kcfx.h (1:1, no logic of its own).
//
// NOT compiled in the build image: it has no JDK (no jni.h, no javac); see INTEGRATION.md for the build line and for
// the Java class (nl.wur.bis.kcftools.Native.Kcfx) whose `native` methods these functions implement. The Java side
// passes direct ByteBuffers / primitive arrays it owns; handles cross as jlong.
#include <jni.h>

#include "../../../include/kcfx.h"

namespace {
void throw_kcfx(JNIEnv* env, int32_t rc) {
    // GetVariants maps any failure to the reference convention (Logger.error -> exit 1, Utils/Logger.java:29-31)
    jclass cls = env->FindClass("java/lang/IllegalStateException");
    if (cls) env->ThrowNew(cls, kcfx_last_error());
    (void)rc;
}
}  // namespace

extern "C" {

// long dbOpen(String kmcPrefix, int device)                              <-> new KMC(prefix, inMemory), KMC.java:56
JNIEXPORT jlong JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_dbOpen(JNIEnv* env, jclass, jstring prefix, jint device) {
    if (!prefix) { throw_kcfx(env, KCFX_ERR_ARG); return 0; }
    const char* p = env->GetStringUTFChars(prefix, nullptr);
    if (!p) return 0;                                       // (OutOfMemoryError is pending)
    kcfx_db* db = nullptr;
    const int32_t rc = kcfx_db_open(p, device, &db);
    env->ReleaseStringUTFChars(prefix, p);
    if (rc) { throw_kcfx(env, rc); return 0; }
    return reinterpret_cast<jlong>(db);
}

// long dbOpenFor(String kmcPrefix, int device, long expectedQueries): the same, the table sized for the look-ups the handle will
// serve (kcfx_db_open_opts) -- getVariations makes ONE pass over its windows per database (GetVariants.java:129-159)
JNIEXPORT jlong JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_dbOpenFor(JNIEnv* env, jclass, jstring prefix, jint device, jlong expectedQueries) {
    if (!prefix) { throw_kcfx(env, KCFX_ERR_ARG); return 0; }
    const char* p = env->GetStringUTFChars(prefix, nullptr);
    kcfx_db* db = nullptr;
    kcfx_db_opts opts;
    opts.expected_queries = expectedQueries;
    opts.no_fallback = 0;
    opts.reserved = 0;
    const int32_t rc = kcfx_db_open_opts(p, device, &opts, &db);
    env->ReleaseStringUTFChars(prefix, p);
    if (rc) { throw_kcfx(env, rc); return 0; }
    return reinterpret_cast<jlong>(db);
}

// int[] dbInfo(long db): {k, mode, counterSize, lutPrefixLength, signatureLength, minCount, maxCount, bothStrands}
JNIEXPORT jintArray JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_dbInfo(JNIEnv* env, jclass, jlong db) {
    kcfx_db_info_t info;
    if (int32_t rc = kcfx_db_info(reinterpret_cast<kcfx_db*>(db), &info)) { throw_kcfx(env, rc); return nullptr; }
    const jint v[8] = {info.kmer_length, info.mode, info.counter_size, info.lut_prefix_length, info.signature_length,
                       info.min_count, info.max_count, info.both_strands};
    jintArray out = env->NewIntArray(8);
    env->SetIntArrayRegion(out, 0, 8, v);
    return out;
}

JNIEXPORT void JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_dbClose(JNIEnv*, jclass, jlong db) {
    kcfx_db_close(reinterpret_cast<kcfx_db*>(db));
}

JNIEXPORT jint JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_releaseCached(JNIEnv*, jclass, jint device) {
    return kcfx_release_cached(device);
}

// double setCacheFraction(double fraction): the share of a device that released buffers may occupy (default 0.5); returns the previous one
JNIEXPORT jdouble JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_setCacheFraction(JNIEnv*, jclass, jdouble fraction) {
    return kcfx_set_cache_fraction(fraction);
}

// void screen(long db, ByteBuffer text, long[] winBegin, int[] winLen, int minKmerCount, ByteBuffer results)
// `text` and `results` are direct buffers; results receives nWindows * 40 bytes (kcfx_result, little-endian).
// winBegin and winLen describe the same windows: non-null, equal lengths (else IllegalArgumentException; the C ABI trusts n)
static bool window_arrays_ok(JNIEnv* env, jlongArray winBegin, jintArray winLen) {
    if (winBegin && winLen && env->GetArrayLength(winBegin) == env->GetArrayLength(winLen)) return true;
    jclass cls = env->FindClass("java/lang/IllegalArgumentException");
    if (cls) env->ThrowNew(cls, "winBegin and winLen must be non-null arrays of the same length");
    return false;
}

JNIEXPORT void JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_screen(JNIEnv* env, jclass, jlong db, jobject text,
                                                                   jlongArray winBegin, jintArray winLen,
                                                                   jint minKmerCount, jobject results) {
    if (!window_arrays_ok(env, winBegin, winLen)) return;
    const auto* t = static_cast<const uint8_t*>(env->GetDirectBufferAddress(text));
    const jlong tlen = env->GetDirectBufferCapacity(text);
    auto* out = static_cast<kcfx_result*>(env->GetDirectBufferAddress(results));
    const jsize n = env->GetArrayLength(winLen);
    if (!t || !out || env->GetDirectBufferCapacity(results) < (jlong)n * (jlong)sizeof(kcfx_result)) {
        jclass cls = env->FindClass("java/lang/IllegalArgumentException");
        if (cls) env->ThrowNew(cls, "text / results must be direct buffers of sufficient capacity");
        return;
    }
    jlong* wb = env->GetLongArrayElements(winBegin, nullptr);
    if (!wb) return;                                        // (OutOfMemoryError is pending)
    jint* wl = env->GetIntArrayElements(winLen, nullptr);
    if (!wl) {
        env->ReleaseLongArrayElements(winBegin, wb, JNI_ABORT);
        return;
    }
    const int32_t rc = kcfx_screen(reinterpret_cast<kcfx_db*>(db), t, tlen, reinterpret_cast<const int64_t*>(wb),
                                   reinterpret_cast<const int32_t*>(wl), n, minKmerCount, out);
    env->ReleaseLongArrayElements(winBegin, wb, JNI_ABORT);
    env->ReleaseIntArrayElements(winLen, wl, JNI_ABORT);
    if (rc) throw_kcfx(env, rc);
}


// shared body of refCreate / refCreateFor: both window arrays must have the same length (the tile planner of
// kcfx_ref_create_opts reads winBegin[i] for every i < winLen.length) and may fail to pin
static jlong ref_create_checked(JNIEnv* env, jint device, jint k, jobject text, jlongArray winBegin, jintArray winLen,
                                const kcfx_ref_opts* opts) {
    auto illegal = [&](const char* msg) -> jlong {
        jclass cls = env->FindClass("java/lang/IllegalArgumentException");
        if (cls) env->ThrowNew(cls, msg);
        return 0;
    };
    if (!text || !winBegin || !winLen) return illegal("text, winBegin and winLen must not be null");
    const auto* t = static_cast<const uint8_t*>(env->GetDirectBufferAddress(text));
    if (!t) return illegal("text must be a direct buffer");
    const jlong tlen = env->GetDirectBufferCapacity(text);
    const jsize n = env->GetArrayLength(winLen);
    if (env->GetArrayLength(winBegin) != n) return illegal("winBegin and winLen differ in length");
    jlong* wb = env->GetLongArrayElements(winBegin, nullptr);
    if (!wb) return 0;                                      // (OutOfMemoryError is pending)
    jint* wl = env->GetIntArrayElements(winLen, nullptr);
    if (!wl) {
        env->ReleaseLongArrayElements(winBegin, wb, JNI_ABORT);
        return 0;
    }
    kcfx_ref* ref = nullptr;
    const int32_t rc = kcfx_ref_create_opts(device, k, t, tlen, reinterpret_cast<const int64_t*>(wb),
                                            reinterpret_cast<const int32_t*>(wl), n, opts, &ref);
    env->ReleaseLongArrayElements(winBegin, wb, JNI_ABORT);
    env->ReleaseIntArrayElements(winLen, wl, JNI_ABORT);
    if (rc) { throw_kcfx(env, rc); return 0; }
    return reinterpret_cast<jlong>(ref);
}

// long refCreate(int device, int k, ByteBuffer text, long[] winBegin, int[] winLen)
JNIEXPORT jlong JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_refCreate(JNIEnv* env, jclass, jint device, jint k, jobject text,
                                                                       jlongArray winBegin, jintArray winLen) {
    return ref_create_checked(env, device, k, text, winBegin, winLen, nullptr);
}

// long refCreateFor(int device, int k, ByteBuffer text, long[] winBegin, int[] winLen, int expectedScreens)
// expectedScreens = 1: ONE database will be screened against this reference (a getVariations run as the plugin stands) -- it is
// kept as text and the screening launch plans its tiles itself (kcfx_ref_create_opts); 0: unknown, planned once for any number
JNIEXPORT jlong JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_refCreateFor(JNIEnv* env, jclass, jint device, jint k, jobject text,
                                                                          jlongArray winBegin, jintArray winLen, jint expectedScreens) {
    kcfx_ref_opts opts;
    opts.expected_screens = expectedScreens;
    opts.reserved = 0;
    return ref_create_checked(env, device, k, text, winBegin, winLen, &opts);
}

// long[] tablePlan(long totalKmers, int k, boolean bothStrands, long freeBytes)  ->  {lines of 128 bytes, regions}; host only
JNIEXPORT jlongArray JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_tablePlan(JNIEnv* env, jclass, jlong totalKmers, jint k,
                                                                            jboolean bothStrands, jlong freeBytes) {
    int64_t lines = 0;
    int32_t regions = 1;
    const int32_t rc = kcfx_table_plan(totalKmers, k, bothStrands ? 1 : 0, freeBytes, &lines, &regions);
    if (rc) { throw_kcfx(env, rc); return nullptr; }
    jlongArray out = env->NewLongArray(2);
    if (!out) return nullptr;
    const jlong v[2] = {(jlong)lines, (jlong)regions};
    env->SetLongArrayRegion(out, 0, 2, v);
    return out;
}

// long[] tablePlanFor(long totalKmers, int k, boolean bothStrands, long freeBytes, long expectedQueries): the table dbOpenFor would build
JNIEXPORT jlongArray JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_tablePlanFor(JNIEnv* env, jclass, jlong totalKmers, jint k,
                                                                               jboolean bothStrands, jlong freeBytes, jlong expectedQueries) {
    int64_t lines = 0;
    int32_t regions = 1;
    const int32_t rc = kcfx_table_plan_for(totalKmers, k, bothStrands ? 1 : 0, freeBytes, expectedQueries, &lines, &regions);
    if (rc) { throw_kcfx(env, rc); return nullptr; }
    jlongArray out = env->NewLongArray(2);
    if (!out) return nullptr;
    const jlong v[2] = {(jlong)lines, (jlong)regions};
    env->SetLongArrayRegion(out, 0, 2, v);
    return out;
}

JNIEXPORT void JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_refDestroy(JNIEnv*, jclass, jlong ref) {
    kcfx_ref_destroy(reinterpret_cast<kcfx_ref*>(ref));
}

// void screenRefMulti(long[] dbs, long ref, int minKmerCount, ByteBuffer results)
// results: dbs.length * nWindows * 40 bytes, sample-major -- what S getVariations runs + cohort produce (Cohort.java:71-119)
JNIEXPORT void JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_screenRefMulti(JNIEnv* env, jclass, jlongArray dbs, jlong ref,
                                                                            jint minKmerCount, jobject results) {
    auto* r = reinterpret_cast<kcfx_ref*>(ref);
    auto* out = static_cast<kcfx_result*>(env->GetDirectBufferAddress(results));
    const jsize s = env->GetArrayLength(dbs);
    const jlong need = (jlong)s * (jlong)kcfx_ref_n_windows(r) * (jlong)sizeof(kcfx_result);
    if (!out || env->GetDirectBufferCapacity(results) < need) {
        jclass cls = env->FindClass("java/lang/IllegalArgumentException");
        if (cls) env->ThrowNew(cls, "results must be a direct buffer of dbs.length * nWindows * 40 bytes");
        return;
    }
    jlong* h = env->GetLongArrayElements(dbs, nullptr);
    static_assert(sizeof(jlong) == sizeof(kcfx_db*), "handles cross as jlong");
    const int32_t rc = kcfx_screen_ref_multi(reinterpret_cast<kcfx_db* const*>(h), s, r, minKmerCount, out);
    env->ReleaseLongArrayElements(dbs, h, JNI_ABORT);
    if (rc) throw_kcfx(env, rc);
}

// long cohortOpen(String[] kmcPrefixes, int device, long expectedQueries) / void cohortClose(long cohort)
// up to 8 samples in ONE table with a counter per sample and cell (kcfx_cohort_open): what S getVariations runs + cohort do
// (run_kcftools.sh:124-160, Cohort.java:71-119), located once per reference k-mer
JNIEXPORT jlong JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_cohortOpen(JNIEnv* env, jclass, jobjectArray prefixes, jint device, jlong expectedQueries) {
    if (!prefixes) { throw_kcfx(env, KCFX_ERR_ARG); return 0; }
    const jsize s = env->GetArrayLength(prefixes);
    for (jsize i = 0; i < s; ++i)
        if (!env->GetObjectArrayElement(prefixes, i)) { throw_kcfx(env, KCFX_ERR_ARG); return 0; }
    const char** names = new const char*[s > 0 ? s : 1];
    for (jsize i = 0; i < s; ++i) names[i] = env->GetStringUTFChars(static_cast<jstring>(env->GetObjectArrayElement(prefixes, i)), nullptr);
    kcfx_db_opts opts;
    opts.expected_queries = expectedQueries;
    opts.no_fallback = 0;
    opts.reserved = 0;
    kcfx_cohort* co = nullptr;
    const int32_t rc = kcfx_cohort_open(names, s, device, &opts, &co);
    for (jsize i = 0; i < s; ++i) env->ReleaseStringUTFChars(static_cast<jstring>(env->GetObjectArrayElement(prefixes, i)), names[i]);
    delete[] names;
    if (rc) { throw_kcfx(env, rc); return 0; }
    return reinterpret_cast<jlong>(co);
}
JNIEXPORT void JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_cohortClose(JNIEnv*, jclass, jlong cohort) {
    kcfx_cohort_close(reinterpret_cast<kcfx_cohort*>(cohort));
}
// void cohortScreenRef(long cohort, long ref, int minKmerCount, ByteBuffer results): results = samples * nWindows * 40 bytes, sample-major
JNIEXPORT void JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_cohortScreenRef(JNIEnv* env, jclass, jlong cohort, jlong ref, jint minKmerCount,
                                                                             jobject results) {
    auto* co = reinterpret_cast<kcfx_cohort*>(cohort);
    auto* r = reinterpret_cast<kcfx_ref*>(ref);
    kcfx_cohort_info_t info;
    if (int32_t rc = kcfx_cohort_info(co, &info)) { throw_kcfx(env, rc); return; }
    auto* out = static_cast<kcfx_result*>(env->GetDirectBufferAddress(results));
    const jlong need = (jlong)info.n_samples * (jlong)kcfx_ref_n_windows(r) * (jlong)sizeof(kcfx_result);
    if (!out || env->GetDirectBufferCapacity(results) < need) {
        jclass cls = env->FindClass("java/lang/IllegalArgumentException");
        if (cls) env->ThrowNew(cls, "results must be a direct buffer of samples * nWindows * 40 bytes");
        return;
    }
    if (int32_t rc = kcfx_cohort_screen_ref(co, r, minKmerCount, out)) throw_kcfx(env, rc);
}

// ---- a table partitioned over devices (kcfx_db_open_slice; include/kcfx.h) -------------------------------------------------------
// long dbOpenSlice(String kmcPrefix, int device, long expectedQueries, int nSlices, int slice)
JNIEXPORT jlong JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_dbOpenSlice(JNIEnv* env, jclass, jstring prefix, jint device, jlong expectedQueries,
                                                                         jint nSlices, jint slice) {
    if (!prefix) { throw_kcfx(env, KCFX_ERR_ARG); return 0; }
    const char* p = env->GetStringUTFChars(prefix, nullptr);
    if (!p) return 0;
    kcfx_db* db = nullptr;
    kcfx_db_opts opts;
    opts.expected_queries = expectedQueries;
    opts.no_fallback = 1;
    opts.reserved = 0;
    const int32_t rc = kcfx_db_open_slice(p, device, &opts, nSlices, slice, &db);
    env->ReleaseStringUTFChars(prefix, p);
    if (rc) { throw_kcfx(env, rc); return 0; }
    return reinterpret_cast<jlong>(db);
}
// void screenRefPartial(long db, long ref, int minKmerCount): this slice's hits into the reference's tile bitmaps (synchronous)
JNIEXPORT void JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_screenRefPartial(JNIEnv* env, jclass, jlong db, jlong ref, jint minKmerCount) {
    if (int32_t rc = kcfx_screen_ref_partial(reinterpret_cast<kcfx_db*>(db), reinterpret_cast<kcfx_ref*>(ref), minKmerCount)) throw_kcfx(env, rc);
}
// void refMergeFrom(long dst, long src): another slice's bitmaps (a reference of the same text and windows, any device) into dst's
JNIEXPORT void JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_refMergeFrom(JNIEnv* env, jclass, jlong dst, jlong src) {
    if (int32_t rc = kcfx_ref_merge_from(reinterpret_cast<kcfx_ref*>(dst), reinterpret_cast<kcfx_ref*>(src))) throw_kcfx(env, rc);
}
// void refFinish(long ref, ByteBuffer results): the window records of what ref's bitmaps hold now; results = nWindows * 40 bytes
JNIEXPORT void JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_refFinish(JNIEnv* env, jclass, jlong ref, jobject results) {
    kcfx_ref* r = reinterpret_cast<kcfx_ref*>(ref);
    auto* out = results ? static_cast<kcfx_result*>(env->GetDirectBufferAddress(results)) : nullptr;
    if (!r || !out || env->GetDirectBufferCapacity(results) < (jlong)kcfx_ref_n_windows(r) * (jlong)sizeof(kcfx_result)) {
        jclass cls = env->FindClass("java/lang/IllegalArgumentException");
        if (cls) env->ThrowNew(cls, "results must be a direct buffer of nWindows * 40 bytes");
        return;
    }
    if (int32_t rc = kcfx_ref_finish(r, out)) throw_kcfx(env, rc);
}

// int[] dbHeader(String kmcPrefix): the dbInfo fields from the header alone, no GPU (KMC.java:107-168; GetVariants.java:96,103)
JNIEXPORT jintArray JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_dbHeader(JNIEnv* env, jclass, jstring prefix) {
    if (!prefix) { throw_kcfx(env, KCFX_ERR_ARG); return nullptr; }
    const char* p = env->GetStringUTFChars(prefix, nullptr);
    if (!p) return nullptr;
    kcfx_db_info_t info;
    const int32_t rc = kcfx_db_header(p, &info);
    env->ReleaseStringUTFChars(prefix, p);
    if (rc) { throw_kcfx(env, rc); return nullptr; }
    const jint v[8] = {info.kmer_length, info.mode, info.counter_size, info.lut_prefix_length, info.signature_length,
                       info.min_count, info.max_count, info.both_strands};
    jintArray out = env->NewIntArray(8);
    env->SetIntArrayRegion(out, 0, 8, v);
    return out;
}

// void screenSharded(int[] devices, String[] kmcPrefixes, int k, ByteBuffer text, long[] winBegin, int[] winLen,
//                    int minKmerCount, int shardMode, ByteBuffer results)
// Every GPU of the node in one call (kcfx_ctx_create + kcfx_screen_sharded): replaces the per-sample processes of
// run_kcftools.sh:124-140 and the thread pool of GetVariants.java:129-151. results: samples * nWindows * 40 bytes, sample-major.
JNIEXPORT void JNICALL Java_nl_wur_bis_kcftools_Native_Kcfx_screenSharded(JNIEnv* env, jclass, jintArray devices, jobjectArray prefixes,
                                                                           jint k, jobject text, jlongArray winBegin, jintArray winLen,
                                                                           jint minKmerCount, jint shardMode, jobject results) {
    if (!window_arrays_ok(env, winBegin, winLen)) return;
    if (!devices || !prefixes) {
        jclass cls = env->FindClass("java/lang/IllegalArgumentException");
        if (cls) env->ThrowNew(cls, "devices and kmcPrefixes must not be null");
        return;
    }
    for (jsize i = 0; i < env->GetArrayLength(prefixes); ++i)
        if (!env->GetObjectArrayElement(prefixes, i)) {
            jclass cls = env->FindClass("java/lang/IllegalArgumentException");
            if (cls) env->ThrowNew(cls, "kmcPrefixes holds a null string");
            return;
        }
    const auto* t = static_cast<const uint8_t*>(env->GetDirectBufferAddress(text));
    auto* out = static_cast<kcfx_result*>(env->GetDirectBufferAddress(results));
    const jsize n = env->GetArrayLength(winLen), s = env->GetArrayLength(prefixes), g = env->GetArrayLength(devices);
    if (!t || !out || env->GetDirectBufferCapacity(results) < (jlong)s * n * (jlong)sizeof(kcfx_result)) {
        jclass cls = env->FindClass("java/lang/IllegalArgumentException");
        if (cls) env->ThrowNew(cls, "text / results must be direct buffers; results holds samples * nWindows * 40 bytes");
        return;
    }
    jint* dev = env->GetIntArrayElements(devices, nullptr);
    kcfx_ctx* ctx = nullptr;
    int32_t rc = kcfx_ctx_create(reinterpret_cast<const int32_t*>(dev), g, &ctx);
    env->ReleaseIntArrayElements(devices, dev, JNI_ABORT);
    if (rc) { throw_kcfx(env, rc); return; }
    const char** names = new const char*[s > 0 ? s : 1];
    for (jsize i = 0; i < s; ++i) names[i] = env->GetStringUTFChars(static_cast<jstring>(env->GetObjectArrayElement(prefixes, i)), nullptr);
    jlong* wb = env->GetLongArrayElements(winBegin, nullptr);
    jint* wl = env->GetIntArrayElements(winLen, nullptr);       // (a null from either: kcfx_screen_sharded returns KCFX_ERR_ARG)
    rc = kcfx_screen_sharded(ctx, names, s, k, t, env->GetDirectBufferCapacity(text), reinterpret_cast<const int64_t*>(wb),
                             reinterpret_cast<const int32_t*>(wl), n, minKmerCount, shardMode, out, nullptr);
    env->ReleaseLongArrayElements(winBegin, wb, JNI_ABORT);
    env->ReleaseIntArrayElements(winLen, wl, JNI_ABORT);
    for (jsize i = 0; i < s; ++i) env->ReleaseStringUTFChars(static_cast<jstring>(env->GetObjectArrayElement(prefixes, i)), names[i]);
    delete[] names;
    kcfx_ctx_destroy(ctx);
    if (rc) throw_kcfx(env, rc);
}

}  // extern "C"
