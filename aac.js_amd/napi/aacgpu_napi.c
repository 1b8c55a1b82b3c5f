/*
 * aacgpu_napi.c — N-API (node_api.h, N-API <= v8) binding of the C ABI in include/aacgpu.h.
 *
 * This is the FFI layer the JavaScript host (aac.js_amd/js) uses in place of the reference's
 * in-process  this.process(elements) + interleave  (src/decoder.js:201-215).  The addon itself is
 * plain C (gcc); it dlopen()s libaacgpu.so, so it builds on machines without hipcc and fails
 * loudly at load time if the HIP library is absent.  Non-zero status codes become thrown Errors,
 * the convention Aurora's Decoder.decode() already expects from readChunk.
 */
#include <node_api.h>

#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../../include/aacgpu.h"

static struct {
    void* dl;
    int  (*create)(const aacg_config*, aacg_engine**);
    void (*destroy)(aacg_engine*);
    const char* (*last_error)(const aacg_engine*);
    int  (*abi_version)(void);
    int  (*reset_stream)(aacg_engine*, uint32_t);
    int  (*get_overlap)(aacg_engine*, uint32_t, uint32_t, float*);
    int  (*set_overlap)(aacg_engine*, uint32_t, uint32_t, const float*);
    int  (*decode_batch)(aacg_engine*, const aacg_unit_desc*, uint32_t, const void*, uint32_t,
                         const aacg_band_meta*, uint32_t, void*, size_t);
    int  (*submit)(aacg_engine*, const aacg_unit_desc*, uint32_t, const void*, uint32_t,
                   const aacg_band_meta*, uint32_t, void*, size_t, uint64_t*);
    int  (*wait)(aacg_engine*, uint64_t);
    int  (*decode_batch_ex)(aacg_engine*, const aacg_batch*);
    int  (*decode_batch_tns)(aacg_engine*, const aacg_unit_desc*, uint32_t, const void*, uint32_t,
                             const aacg_band_meta*, uint32_t, const aacg_tns_info*, uint32_t, void*, size_t);
    int  (*submit_tns)(aacg_engine*, const aacg_unit_desc*, uint32_t, const void*, uint32_t,
                       const aacg_band_meta*, uint32_t, const aacg_tns_info*, uint32_t, void*, size_t, uint64_t*);
    int  (*parser_create)(int, int, const aacg_code_entry*, const uint32_t*, aacg_parser**);
    void (*parser_destroy)(aacg_parser*);
    const char* (*parser_last_error)(const aacg_parser*);
    const char* (*parse_status_string)(int);
    int  (*parse_batch)(aacg_parser*, const uint8_t*, size_t, const aacg_parse_frame*, uint32_t, uint32_t, uint32_t, uint32_t,
                        aacg_unit_desc*, int16_t*, aacg_band_meta*, aacg_tns_info*, aacg_parse_result*);
    int  (*pipeline_create)(const aacg_pipeline_config*, const aacg_code_entry*, const uint32_t*, aacg_pipeline**);
    void (*pipeline_destroy)(aacg_pipeline*);
    const char* (*pipeline_last_error)(const aacg_pipeline*);
    int  (*pipeline_reset_stream)(aacg_pipeline*, uint32_t);
    int  (*pipeline_decode)(aacg_pipeline*, const uint8_t*, size_t, const aacg_parse_frame*, const uint32_t*, uint32_t, uint32_t,
                            void*, aacg_parse_result*, uint32_t*);
    int  (*pipeline_submit)(aacg_pipeline*, const uint8_t*, size_t, const aacg_parse_frame*, const uint32_t*, uint32_t, uint32_t,
                            void*, aacg_parse_result*, uint32_t*, uint64_t*);
    int  (*pipeline_collect)(aacg_pipeline*, uint64_t);
    void* (*host_alloc)(size_t);
    void (*host_free)(void*);
} L;

#define CHECK(env, call) do { if ((call) != napi_ok) { napi_throw_error((env), NULL, "aacgpu: N-API call failed: " #call); return NULL; } } while (0)

static napi_value fail(napi_env env, aacg_engine* e, int rc, const char* what)
{
    char msg[1024];
    snprintf(msg, sizeof msg, "aacgpu: %s failed (%d): %s", what, rc, e && L.last_error ? L.last_error(e) : "");
    napi_throw_error(env, NULL, msg);
    return NULL;
}

static int load_lib(napi_env env, const char* path)
{
    if (L.dl) return 1;
    L.dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!L.dl) {
        char msg[1024];
        snprintf(msg, sizeof msg, "aacgpu: cannot load %s: %s (build it with hipcc --offload-arch=gfx950; there is no CPU fallback)", path, dlerror());
        napi_throw_error(env, NULL, msg);
        return 0;
    }
#define SYM(field, name) do { *(void**)&L.field = dlsym(L.dl, name); if (!L.field) { napi_throw_error(env, NULL, "aacgpu: missing symbol " name); return 0; } } while (0)
    SYM(create, "aacg_create"); SYM(destroy, "aacg_destroy"); SYM(last_error, "aacg_last_error");
    SYM(abi_version, "aacg_abi_version"); SYM(reset_stream, "aacg_reset_stream");
    SYM(get_overlap, "aacg_get_overlap"); SYM(set_overlap, "aacg_set_overlap"); SYM(decode_batch, "aacg_decode_batch");
    SYM(submit, "aacg_submit"); SYM(wait, "aacg_wait");
    SYM(decode_batch_ex, "aacg_decode_batch_ex"); SYM(decode_batch_tns, "aacg_decode_batch_tns"); SYM(submit_tns, "aacg_submit_tns");
    SYM(parser_create, "aacg_parser_create"); SYM(parser_destroy, "aacg_parser_destroy"); SYM(parser_last_error, "aacg_parser_last_error");
    SYM(parse_status_string, "aacg_parse_status_string"); SYM(parse_batch, "aacg_parse_batch");
    SYM(pipeline_create, "aacg_pipeline_create"); SYM(pipeline_destroy, "aacg_pipeline_destroy"); SYM(pipeline_last_error, "aacg_pipeline_last_error");
    SYM(pipeline_reset_stream, "aacg_pipeline_reset_stream"); SYM(pipeline_decode, "aacg_pipeline_decode");
    SYM(pipeline_submit, "aacg_pipeline_submit"); SYM(pipeline_collect, "aacg_pipeline_collect");
    SYM(host_alloc, "aacg_host_alloc"); SYM(host_free, "aacg_host_free");
#undef SYM
    return 1;
}

static int get_i32(napi_env env, napi_value obj, const char* key, int32_t dflt)
{
    napi_value v; bool has = false; int32_t out = dflt;
    if (napi_has_named_property(env, obj, key, &has) == napi_ok && has &&
        napi_get_named_property(env, obj, key, &v) == napi_ok) napi_get_value_int32(env, v, &out);
    return out;
}

/* What an external points to: the native object behind a tag that says which kind it is (a parser handle passed where an
 * engine is expected is refused instead of being cast), and a lock: an aacg_engine is not re-entrant (include/aacgpu.h:
 * externally serialised), so synchronous calls and the async jobs of one engine — which run on libuv threads — take
 * turns.  Jobs keep a reference on the external, so a collected Engine object cannot be finalised under a running job. */
#define BOX_ENGINE 0x41454e47u   /* 'AENG' */
#define BOX_PARSER 0x41505253u   /* 'APRS' */
#define BOX_PIPELINE 0x4150504cu /* 'APPL' */
#define PCM_RING_MAX 16
typedef struct { uint32_t kind; void* ptr; pthread_mutex_t lock; int out_i16; /* engine / pipeline: AACG_OUTPUT_I16 */
                 /* pipeline, { pcmRing: K }: K page-locked PCM buffers made once and handed out in turn (pipelineDecode) */
                 napi_ref ring_ab[PCM_RING_MAX]; void* ring_ptr[PCM_RING_MAX]; size_t ring_bytes; int ring_n; unsigned ring_next;
                 /* pipeline: the batches submitted and not yet collected, oldest first (pipelineSubmit / pipelineCollect) */
                 void* jobs[4]; int n_jobs; } handle_box;

static handle_box* box_new(uint32_t kind, void* ptr)
{
    handle_box* b = (handle_box*)calloc(1, sizeof *b);
    if (!b) return NULL;
    b->kind = kind; b->ptr = ptr;
    pthread_mutex_init(&b->lock, NULL);
    return b;
}

static handle_box* box_of(napi_env env, napi_value v, uint32_t kind, const char* what)
{
    void* p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p || ((handle_box*)p)->kind != kind) { napi_throw_error(env, NULL, what); return NULL; }
    return (handle_box*)p;
}

static void engine_finalize(napi_env env, void* data, void* hint)
{
    (void)env; (void)hint;
    handle_box* b = (handle_box*)data;
    if (!b) return;
    if (b->ptr && L.destroy) L.destroy((aacg_engine*)b->ptr);
    b->kind = 0;
    pthread_mutex_destroy(&b->lock);
    free(b);
}

/* load(path) -> abi version */
static napi_value js_load(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1]; char path[2048]; size_t n = 0; napi_value out;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    CHECK(env, napi_get_value_string_utf8(env, argv[0], path, sizeof path, &n));
    if (!load_lib(env, path)) return NULL;
    CHECK(env, napi_create_int32(env, L.abi_version(), &out));
    return out;
}

/* create({deviceOrdinal, sampleIndex, maxStreams, maxChannels, maxBatchUnits, inputKind}) -> external */
static napi_value js_create(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1], out;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!L.dl) { napi_throw_error(env, NULL, "aacgpu: call load(path) first"); return NULL; }
    aacg_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = AACG_ABI_VERSION;
    cfg.device_ordinal = get_i32(env, argv[0], "deviceOrdinal", 0);
    cfg.sample_index = get_i32(env, argv[0], "sampleIndex", 3);
    cfg.max_streams = get_i32(env, argv[0], "maxStreams", 1);
    cfg.max_channels = get_i32(env, argv[0], "maxChannels", 2);
    cfg.max_batch_units = get_i32(env, argv[0], "maxBatchUnits", 0);
    cfg.input_kind = get_i32(env, argv[0], "inputKind", AACG_INPUT_QUANT_I16);
    cfg.tns_mode = get_i32(env, argv[0], "tnsMode", AACG_TNS_REFERENCE);
    cfg.pns_mode = get_i32(env, argv[0], "pnsMode", AACG_PNS_REFERENCE);
    cfg.output_kind = get_i32(env, argv[0], "outputKind", AACG_OUTPUT_F32);
    cfg.cce_mode = get_i32(env, argv[0], "cceMode", AACG_CCE_REFERENCE);
    aacg_engine* e = NULL;
    int rc = L.create(&cfg, &e);
    if (rc) return fail(env, NULL, rc, "aacg_create (is a GPU visible?)");
    handle_box* b = box_new(BOX_ENGINE, e);
    if (!b) { L.destroy(e); napi_throw_error(env, NULL, "aacgpu: out of memory"); return NULL; }
    b->out_i16 = cfg.output_kind == AACG_OUTPUT_I16;
    CHECK(env, napi_create_external(env, b, engine_finalize, NULL, &out));
    return out;
}

static handle_box* engine_box(napi_env env, napi_value v) { return box_of(env, v, BOX_ENGINE, "aacgpu: bad engine handle"); }
/* The synchronous entry points hold the engine's lock for the whole library call: an async job of the same engine —
 * queued or running on a libuv thread — takes the same lock around aacg_submit / aacg_wait, so the two can never be
 * inside the (non-re-entrant) engine at once.  (The JS wrapper awaits its jobs anyway; this is the backstop.) */
static aacg_engine* engine_lock(napi_env env, napi_value v, handle_box** box)
{
    handle_box* b = engine_box(env, v);
    *box = b;
    if (!b) return NULL;
    pthread_mutex_lock(&b->lock);
    return (aacg_engine*)b->ptr;
}
static void engine_unlock(handle_box* b) { pthread_mutex_unlock(&b->lock); }

static int typed(napi_env env, napi_value v, napi_typedarray_type* type, size_t* len, void** data)
{
    napi_value ab; size_t off;
    return napi_get_typedarray_info(env, v, type, len, data, &ab, &off) == napi_ok;
}

/* optional trailing argument: tns:Uint8Array of aacg_tns_info records (AACG_TNS_SPEC engines); 1 ok, 0 thrown */
static int optional_tns(napi_env env, size_t argc, napi_value* argv, size_t at, void** data, size_t* count)
{
    *data = NULL; *count = 0;
    if (argc <= at) return 1;
    napi_valuetype vt;
    if (napi_typeof(env, argv[at], &vt) != napi_ok) return 0;
    if (vt == napi_null || vt == napi_undefined) return 1;
    napi_typedarray_type tt; size_t nt;
    if (!typed(env, argv[at], &tt, &nt, data) || tt != napi_uint8_array || nt % sizeof(aacg_tns_info)) {
        napi_throw_type_error(env, NULL, "tns must be a Uint8Array of 424-byte aacg_tns_info records");
        return 0;
    }
    *count = nt / sizeof(aacg_tns_info);
    return 1;
}

/* optional trailing argument: cce:Uint8Array of aacg_cce_info records (AACG_CCE_SPEC engines); 1 ok, 0 thrown */
static int optional_cce(napi_env env, size_t argc, napi_value* argv, size_t at, void** data, size_t* count)
{
    *data = NULL; *count = 0;
    if (argc <= at) return 1;
    napi_valuetype vt;
    if (napi_typeof(env, argv[at], &vt) != napi_ok) return 0;
    if (vt == napi_null || vt == napi_undefined) return 1;
    napi_typedarray_type tt; size_t nt;
    if (!typed(env, argv[at], &tt, &nt, data) || tt != napi_uint8_array || nt % sizeof(aacg_cce_info)) {
        napi_throw_type_error(env, NULL, "cce must be a Uint8Array of 7716-byte aacg_cce_info records");
        return 0;
    }
    *count = nt / sizeof(aacg_cce_info);
    return 1;
}

/* decodeBatch(engine, units:Uint8Array(64*n), coeffs:Int16Array|Float32Array, meta:Uint16Array|null, pcm:Float32Array
 *             [, tns:Uint8Array(424*m) [, cce:Uint8Array(7716*k)]]) */
static napi_value js_decode_batch(napi_env env, napi_callback_info info)
{
    size_t argc = 7; napi_value argv[7];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void* dt; size_t nt; void* dcce; size_t ncce;
    if (!optional_tns(env, argc, argv, 5, &dt, &nt) || !optional_cce(env, argc, argv, 6, &dcce, &ncce)) return NULL;
    handle_box* box = engine_box(env, argv[0]);
    if (!box) return NULL;
    napi_typedarray_type tu, tc, tm, tp; size_t nu, nc, nm = 0, np; void *du, *dc, *dm = NULL, *dp;
    if (!typed(env, argv[1], &tu, &nu, &du) || tu != napi_uint8_array || nu % sizeof(aacg_unit_desc)) {
        napi_throw_type_error(env, NULL, "units must be a Uint8Array of 64-byte aacg_unit_desc records"); return NULL; }
    if (!typed(env, argv[2], &tc, &nc, &dc) || (tc != napi_int16_array && tc != napi_float32_array) || nc % 1024) {
        napi_throw_type_error(env, NULL, "coeffs must be an Int16Array or Float32Array, a multiple of 1024 long"); return NULL; }
    napi_valuetype vt;
    CHECK(env, napi_typeof(env, argv[3], &vt));
    if (vt != napi_null && vt != napi_undefined) {
        if (!typed(env, argv[3], &tm, &nm, &dm) || tm != napi_uint16_array || nm % AACG_MAX_SECTIONS) {
            napi_throw_type_error(env, NULL, "meta must be a Uint16Array of 120-word aacg_band_meta records"); return NULL; }
    }
    if (!typed(env, argv[4], &tp, &np, &dp) || tp != (box->out_i16 ? napi_int16_array : napi_float32_array)) {
        napi_throw_type_error(env, NULL, "pcm must be a Float32Array (an Int16Array for an engine created with outputKind: OUTPUT_I16)"); return NULL; }
    aacg_batch b;
    memset(&b, 0, sizeof b);
    b.units = (const aacg_unit_desc*)du; b.n_units = (uint32_t)(nu / sizeof(aacg_unit_desc));
    b.coeffs = dc; b.n_coef_blocks = (uint32_t)(nc / 1024);
    b.meta = (const aacg_band_meta*)dm; b.n_meta = (uint32_t)(nm / AACG_MAX_SECTIONS);
    b.tns = (const aacg_tns_info*)dt; b.n_tns = (uint32_t)nt;
    b.cce = (const aacg_cce_info*)dcce; b.n_cce = (uint32_t)ncce;
    b.pcm_out = dp; b.n_pcm_floats = np;
    aacg_engine* e = engine_lock(env, argv[0], &box);
    if (!e) return NULL;
    int rc = L.decode_batch_ex(e, &b);
    napi_value failed = rc ? fail(env, e, rc, "aacg_decode_batch") : NULL;     /* reads aacg_last_error: still under the lock */
    engine_unlock(box);
    if (rc) return failed;
    return argv[4];
}

static napi_value js_reset_stream(napi_env env, napi_callback_info info)
{
    size_t argc = 2; napi_value argv[2]; uint32_t s = 0;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_box* box;
    napi_get_value_uint32(env, argv[1], &s);
    aacg_engine* e = engine_lock(env, argv[0], &box);
    if (!e) return NULL;
    int rc = L.reset_stream(e, s);
    if (rc) (void)fail(env, e, rc, "aacg_reset_stream");
    engine_unlock(box);
    return NULL;
}

/* getOverlap(engine, stream, channel, Float32Array(1024)) / setOverlap(...) */
static napi_value overlap_io(napi_env env, napi_callback_info info, int set)
{
    size_t argc = 4; napi_value argv[4]; uint32_t s = 0, c = 0;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_box* box;
    napi_get_value_uint32(env, argv[1], &s);
    napi_get_value_uint32(env, argv[2], &c);
    napi_typedarray_type t; size_t n; void* d;
    if (!typed(env, argv[3], &t, &n, &d) || t != napi_float32_array || n != 1024) {
        napi_throw_type_error(env, NULL, "overlap buffer must be a Float32Array(1024)"); return NULL; }
    aacg_engine* e = engine_lock(env, argv[0], &box);
    if (!e) return NULL;
    int rc = set ? L.set_overlap(e, s, c, (const float*)d) : L.get_overlap(e, s, c, (float*)d);
    if (rc) (void)fail(env, e, rc, set ? "aacg_set_overlap" : "aacg_get_overlap");
    engine_unlock(box);
    return rc ? NULL : argv[3];
}
static napi_value js_get_overlap(napi_env env, napi_callback_info info) { return overlap_io(env, info, 0); }
static napi_value js_set_overlap(napi_env env, napi_callback_info info) { return overlap_io(env, info, 1); }

/* ---- decodeBatchAsync: the same call off the JavaScript thread (napi_async_work), so the event loop keeps
 * parsing the next frames while the GPU works; resolves with the pcm array, rejects with an Error ---------- */
typedef struct {
    napi_async_work work;
    napi_deferred deferred;
    napi_ref refs[6];            /* units, coeffs, meta, pcm, tns and the engine handle stay alive until completion */
    handle_box* box;
    const aacg_tns_info* tns; uint32_t n_tns;
    aacg_engine* e;
    const aacg_unit_desc* units; uint32_t n_units;
    const void* coeffs; uint32_t n_blocks;
    const aacg_band_meta* meta; uint32_t n_meta;
    void* pcm; size_t n_pcm;
    int rc;
    char err[512];
} async_job;

static void job_execute(napi_env env, void* data)
{
    (void)env;
    async_job* j = (async_job*)data;
    uint64_t t = 0;
    pthread_mutex_lock(&j->box->lock);             /* jobs of one engine run one at a time, whatever thread they land on */
    j->rc = L.submit_tns(j->e, j->units, j->n_units, j->coeffs, j->n_blocks, j->meta, j->n_meta, j->tns, j->n_tns,
                         j->pcm, j->n_pcm, &t);
    if (!j->rc) j->rc = L.wait(j->e, t);
    if (j->rc) snprintf(j->err, sizeof j->err, "aacgpu: decodeBatchAsync failed (%d): %.400s", j->rc, L.last_error(j->e));
    pthread_mutex_unlock(&j->box->lock);
}

static void job_complete(napi_env env, napi_status status, void* data)
{
    async_job* j = (async_job*)data;
    napi_value v;
    if (status == napi_ok && j->rc == 0) {
        napi_get_reference_value(env, j->refs[3], &v);
        napi_resolve_deferred(env, j->deferred, v);
    } else {
        napi_value msg;
        napi_create_string_utf8(env, j->rc ? j->err : "aacgpu: async work cancelled", NAPI_AUTO_LENGTH, &msg);
        napi_create_error(env, NULL, msg, &v);
        napi_reject_deferred(env, j->deferred, v);
    }
    for (int i = 0; i < 6; i++) if (j->refs[i]) napi_delete_reference(env, j->refs[i]);
    napi_delete_async_work(env, j->work);
    free(j);
}

static napi_value js_decode_batch_async(napi_env env, napi_callback_info info)
{
    size_t argc = 6; napi_value argv[6], promise, name;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    void* dt; size_t nt;
    if (!optional_tns(env, argc, argv, 5, &dt, &nt)) return NULL;
    handle_box* box = engine_box(env, argv[0]);
    if (!box) return NULL;
    aacg_engine* e = (aacg_engine*)box->ptr;
    napi_typedarray_type tu, tc, tm, tp; size_t nu, nc, nm = 0, np; void *du, *dc, *dm = NULL, *dp;
    napi_valuetype vt;
    if (!typed(env, argv[1], &tu, &nu, &du) || tu != napi_uint8_array || nu % sizeof(aacg_unit_desc) ||
        !typed(env, argv[2], &tc, &nc, &dc) || (tc != napi_int16_array && tc != napi_float32_array) || nc % 1024 ||
        !typed(env, argv[4], &tp, &np, &dp) || tp != (box->out_i16 ? napi_int16_array : napi_float32_array)) {
        napi_throw_type_error(env, NULL, "decodeBatchAsync(engine, Uint8Array units, Int16Array|Float32Array coeffs, Uint16Array|null meta, Float32Array pcm)");
        return NULL;
    }
    CHECK(env, napi_typeof(env, argv[3], &vt));
    if (vt != napi_null && vt != napi_undefined &&
        (!typed(env, argv[3], &tm, &nm, &dm) || tm != napi_uint16_array || nm % AACG_MAX_SECTIONS)) {
        napi_throw_type_error(env, NULL, "meta must be a Uint16Array of 120-word aacg_band_meta records");
        return NULL;
    }
    async_job* j = (async_job*)calloc(1, sizeof *j);
    if (!j) { napi_throw_error(env, NULL, "aacgpu: out of memory"); return NULL; }
    j->e = e; j->box = box; j->units = (const aacg_unit_desc*)du;
    napi_create_reference(env, argv[0], 1, &j->refs[5]); j->n_units = (uint32_t)(nu / sizeof(aacg_unit_desc));
    j->coeffs = dc; j->n_blocks = (uint32_t)(nc / 1024); j->meta = (const aacg_band_meta*)dm; j->n_meta = (uint32_t)(nm / AACG_MAX_SECTIONS);
    j->pcm = dp; j->n_pcm = np;
    j->tns = (const aacg_tns_info*)dt; j->n_tns = (uint32_t)nt;
    if (dt) napi_create_reference(env, argv[5], 1, &j->refs[4]);
    const int idx[4] = {1, 2, 3, 4};
    for (int i = 0; i < 4; i++)
        if (!(i == 2 && !dm)) napi_create_reference(env, argv[idx[i]], 1, &j->refs[i]);
    CHECK(env, napi_create_promise(env, &j->deferred, &promise));
    CHECK(env, napi_create_string_utf8(env, "aacgpu.decodeBatchAsync", NAPI_AUTO_LENGTH, &name));
    CHECK(env, napi_create_async_work(env, NULL, name, job_execute, job_complete, j, &j->work));
    CHECK(env, napi_queue_async_work(env, j->work));
    return promise;
}

/* ---- device front end (aacg_parser_*) ------------------------------------------------------------------ */
static void parser_finalize(napi_env env, void* data, void* hint)
{
    (void)env; (void)hint;
    handle_box* b = (handle_box*)data;
    if (!b) return;
    if (b->ptr && L.parser_destroy) L.parser_destroy((aacg_parser*)b->ptr);
    b->kind = 0;
    pthread_mutex_destroy(&b->lock);
    free(b);
}

/* parserCreate({deviceOrdinal, sampleIndex}, entries:Uint8Array(12*n) of aacg_code_entry, counts:Uint32Array(12)) -> external */
static napi_value js_parser_create(napi_env env, napi_callback_info info)
{
    size_t argc = 3; napi_value argv[3], out;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!L.dl) { napi_throw_error(env, NULL, "aacgpu: call load(path) first"); return NULL; }
    napi_typedarray_type te, tc; size_t ne, nc; void *de, *dc;
    if (!typed(env, argv[1], &te, &ne, &de) || te != napi_uint8_array || ne % sizeof(aacg_code_entry) ||
        !typed(env, argv[2], &tc, &nc, &dc) || tc != napi_uint32_array || nc != 12) {
        napi_throw_type_error(env, NULL, "parserCreate(opts, entries:Uint8Array of 12-byte aacg_code_entry, counts:Uint32Array(12))"); return NULL; }
    size_t total = 0;
    for (int b = 0; b < 12; b++) total += ((const uint32_t*)dc)[b];
    if (total * sizeof(aacg_code_entry) != ne) { napi_throw_type_error(env, NULL, "counts do not add up to the number of entries"); return NULL; }
    aacg_parser* p = NULL;
    int rc = L.parser_create(get_i32(env, argv[0], "deviceOrdinal", 0), get_i32(env, argv[0], "sampleIndex", 3),
                             (const aacg_code_entry*)de, (const uint32_t*)dc, &p);
    if (rc) {
        char msg[1024];
        snprintf(msg, sizeof msg, "aacgpu: aacg_parser_create failed (%d): %s", rc, p ? L.parser_last_error(p) : "");
        if (p) L.parser_destroy(p);
        napi_throw_error(env, NULL, msg);
        return NULL;
    }
    handle_box* b = box_new(BOX_PARSER, p);
    if (!b) { L.parser_destroy(p); napi_throw_error(env, NULL, "aacgpu: out of memory"); return NULL; }
    CHECK(env, napi_create_external(env, b, parser_finalize, NULL, &out));
    return out;
}

/* parseBatch(parser, bytes:Uint8Array, frames:Uint32Array(2*n) [offset, length]..., maxUnits, maxChannels, options,
 *            units:Uint8Array(64*n*maxUnits), q:Int16Array(1024*n*maxChannels), meta:Uint16Array(120*n*maxChannels),
 *            tns:Uint8Array(424*n*maxChannels)|null, results:Uint8Array(8*n)) */
static napi_value js_parse_batch(napi_env env, napi_callback_info info)
{
    size_t argc = 11; napi_value argv[11];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_box* pb = argc < 11 ? NULL : box_of(env, argv[0], BOX_PARSER, "aacgpu: bad parser handle");
    if (argc < 11) napi_throw_error(env, NULL, "aacgpu: parseBatch takes 11 arguments");
    if (!pb) return NULL;
    aacg_parser* p = (aacg_parser*)pb->ptr;
    napi_typedarray_type t; size_t nb, nf, nu, nq, nm, nr; void *db, *df, *du, *dq, *dm, *dr, *dt; size_t nt;
    uint32_t max_units = 0, max_ch = 0, options = 0;
    napi_get_value_uint32(env, argv[3], &max_units); napi_get_value_uint32(env, argv[4], &max_ch); napi_get_value_uint32(env, argv[5], &options);
    if (!typed(env, argv[1], &t, &nb, &db) || t != napi_uint8_array || !typed(env, argv[2], &t, &nf, &df) || t != napi_uint32_array || (nf & 1)) {
        napi_throw_type_error(env, NULL, "parseBatch: bytes must be a Uint8Array, frames a Uint32Array of (offset, length) pairs"); return NULL; }
    const size_t n = nf / 2;
    if (!typed(env, argv[6], &t, &nu, &du) || t != napi_uint8_array || nu != n * max_units * sizeof(aacg_unit_desc) ||
        !typed(env, argv[7], &t, &nq, &dq) || t != napi_int16_array || nq != n * max_ch * 1024 ||
        !typed(env, argv[8], &t, &nm, &dm) || t != napi_uint16_array || nm != n * max_ch * AACG_MAX_SECTIONS ||
        !typed(env, argv[10], &t, &nr, &dr) || t != napi_uint8_array || nr != n * sizeof(aacg_parse_result)) {
        napi_throw_type_error(env, NULL, "parseBatch: output arrays do not have the sizes n * maxUnits / maxChannels imply"); return NULL; }
    if (!optional_tns(env, argc, argv, 9, &dt, &nt)) return NULL;
    if (dt && nt != n * max_ch) { napi_throw_type_error(env, NULL, "parseBatch: tns must hold n * maxChannels records"); return NULL; }
    int rc = L.parse_batch(p, (const uint8_t*)db, nb, (const aacg_parse_frame*)df, (uint32_t)n, max_units, max_ch, options,
                           (aacg_unit_desc*)du, (int16_t*)dq, (aacg_band_meta*)dm, (aacg_tns_info*)dt, (aacg_parse_result*)dr);
    if (rc) {
        char msg[1024];
        snprintf(msg, sizeof msg, "aacgpu: aacg_parse_batch failed (%d): %s", rc, L.parser_last_error(p));
        napi_throw_error(env, NULL, msg);
        return NULL;
    }
    return argv[10];
}

/* ---- bytes in, PCM out (aacg_pipeline_*): the resident route behind SharedEngine({ resident: true }) ------------------- */
static void ring_finalize(napi_env env, void* data, void* hint) { (void)env; (void)hint; if (data && L.host_free) L.host_free(data); }

static void pipe_abandon(napi_env env, void* job);    /* a submitted batch nobody collected: wait for it, drop it */
static void pipeline_finalize(napi_env env, void* data, void* hint)
{
    (void)hint;
    handle_box* b = (handle_box*)data;
    if (!b) return;
    for (int i = 0; i < b->n_jobs; i++) pipe_abandon(env, b->jobs[i]);
    b->n_jobs = 0;
    if (b->ptr && L.pipeline_destroy) L.pipeline_destroy((aacg_pipeline*)b->ptr);
    for (int i = 0; i < b->ring_n; i++) if (b->ring_ab[i]) napi_delete_reference(env, b->ring_ab[i]);   /* (the buffers go with their ArrayBuffers: ring_finalize) */
    b->kind = 0;
    pthread_mutex_destroy(&b->lock);
    free(b);
}

/* pipelineCreate({deviceOrdinal, sampleIndex, maxStreams, channels, maxFrames, outputKind, parseOptions}, entries, counts) -> external */
static napi_value js_pipeline_create(napi_env env, napi_callback_info info)
{
    size_t argc = 3; napi_value argv[3], out;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!L.dl) { napi_throw_error(env, NULL, "aacgpu: call load(path) first"); return NULL; }
    napi_typedarray_type te, tc; size_t ne, nc; void *de, *dc;
    if (!typed(env, argv[1], &te, &ne, &de) || te != napi_uint8_array || ne % sizeof(aacg_code_entry) ||
        !typed(env, argv[2], &tc, &nc, &dc) || tc != napi_uint32_array || nc != 12) {
        napi_throw_type_error(env, NULL, "pipelineCreate(opts, entries:Uint8Array of 12-byte aacg_code_entry, counts:Uint32Array(12))"); return NULL; }
    size_t total = 0;
    for (int b = 0; b < 12; b++) total += ((const uint32_t*)dc)[b];
    if (total * sizeof(aacg_code_entry) != ne) { napi_throw_type_error(env, NULL, "counts do not add up to the number of entries"); return NULL; }
    aacg_pipeline_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = AACG_ABI_VERSION;
    cfg.device_ordinal = get_i32(env, argv[0], "deviceOrdinal", 0);
    cfg.sample_index = get_i32(env, argv[0], "sampleIndex", 3);
    cfg.max_streams = get_i32(env, argv[0], "maxStreams", 1);
    cfg.channels = get_i32(env, argv[0], "channels", 2);
    cfg.max_frames = get_i32(env, argv[0], "maxFrames", 16);
    cfg.output_kind = get_i32(env, argv[0], "outputKind", AACG_OUTPUT_F32);
    cfg.parse_options = get_i32(env, argv[0], "parseOptions", AACG_PARSE_REFERENCE_QUIRKS);
    cfg.lanes = get_i32(env, argv[0], "lanes", 0);
    aacg_pipeline* p = NULL;
    int rc = L.pipeline_create(&cfg, (const aacg_code_entry*)de, (const uint32_t*)dc, &p);
    if (rc) {
        char msg[256];
        snprintf(msg, sizeof msg, "aacgpu: aacg_pipeline_create failed (%d) (is a GPU visible?)", rc);
        napi_throw_error(env, NULL, msg);
        return NULL;
    }
    handle_box* b = box_new(BOX_PIPELINE, p);
    if (!b) { L.pipeline_destroy(p); napi_throw_error(env, NULL, "aacgpu: out of memory"); return NULL; }
    b->out_i16 = cfg.output_kind == AACG_OUTPUT_I16;
    CHECK(env, napi_create_external(env, b, pipeline_finalize, NULL, &out));
    return out;
}

/* Page-locked PCM buffers handed to JavaScript as external ArrayBuffers: the PCM of a batch comes down from the device
 * straight into the memory the Float32Array views — no staging copy, no zero-filled allocation of tens of megabytes per batch.
 * A buffer returns to the pool when the garbage collector has let go of the batch's last frame (the finalizer runs on the
 * JavaScript thread, like every call into this addon); the pool holds a few, the rest are freed.
 * Finalizers do not run inside a synchronous loop, though, and that is how a host drains 256 streams: every flush would then
 * page-lock a fresh 33 MB (3.5-6 ms with hipHostMalloc; 0.6 since round 6: aacg_host_alloc).  So a helper thread keeps the NEXT buffers ready: when
 * one is taken, it makes another of that size while JavaScript slices the batch, and it does the freeing too. */
#define PCM_POOL_MAX 8
#define PCM_READY_MAX 3
typedef struct { void* ptr; size_t bytes; } pcm_buf;
static pcm_buf g_pool[PCM_POOL_MAX];               /* recycled by finalizers (the JavaScript thread of whichever environment — main or a
                                                      worker — let go of the buffer: under g_pool_lock) */
static int g_pool_n = 0;
static pthread_mutex_t g_pool_lock = PTHREAD_MUTEX_INITIALIZER;
static int pool_put(void* p, size_t bytes)          /* 1: kept for the next batch of that size */
{
    int kept = 0;
    pthread_mutex_lock(&g_pool_lock);
    if (g_pool_n < PCM_POOL_MAX) { g_pool[g_pool_n].ptr = p; g_pool[g_pool_n].bytes = bytes; g_pool_n++; kept = 1; }
    pthread_mutex_unlock(&g_pool_lock);
    return kept;
}
static pthread_mutex_t g_prep_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_prep_wake = PTHREAD_COND_INITIALIZER;
static pthread_cond_t g_ready_wake = PTHREAD_COND_INITIALIZER;   /* the helper has put a buffer into g_ready */
static pcm_buf g_ready[PCM_READY_MAX];             /* made ahead by the helper (under g_prep_lock) */
static int g_ready_n = 0;
static size_t g_want_bytes = 0;                    /* the size the helper keeps ready */
static void* g_trash[64];                          /* buffers for the helper to free */
static int g_trash_n = 0;
static int g_prep_started = 0;

static void* prep_main(void* arg)
{
    (void)arg;
    pthread_mutex_lock(&g_prep_lock);
    for (;;) {
        if (g_trash_n) {
            void* p = g_trash[--g_trash_n];
            pthread_mutex_unlock(&g_prep_lock);
            L.host_free(p);
            pthread_mutex_lock(&g_prep_lock);
            continue;
        }
        if (g_want_bytes && g_ready_n < PCM_READY_MAX) {
            const size_t bytes = g_want_bytes;
            pthread_mutex_unlock(&g_prep_lock);
            void* p = L.host_alloc(bytes);
            pthread_mutex_lock(&g_prep_lock);
            if (p) {
                if (g_ready_n < PCM_READY_MAX && bytes == g_want_bytes) { g_ready[g_ready_n].ptr = p; g_ready[g_ready_n].bytes = bytes; g_ready_n++; pthread_cond_broadcast(&g_ready_wake); }
                else if (g_trash_n < 64) g_trash[g_trash_n++] = p;
            } else { g_want_bytes = 0; pthread_cond_broadcast(&g_ready_wake); }      /* out of page-locked memory: the caller's own attempt will report it */
            continue;
        }
        pthread_cond_wait(&g_prep_wake, &g_prep_lock);
    }
    return NULL;
}

static void pcm_discard(void* p)                   /* free off the JavaScript thread where possible */
{
    pthread_mutex_lock(&g_prep_lock);
    const int queued = g_prep_started && g_trash_n < 64;
    if (queued) { g_trash[g_trash_n++] = p; pthread_cond_signal(&g_prep_wake); }
    pthread_mutex_unlock(&g_prep_lock);
    if (!queued) L.host_free(p);
}

static void pcm_finalize(napi_env env, void* data, void* hint)
{
    const size_t bytes = (size_t)hint;
    (void)env;
    if (!pool_put(data, bytes)) pcm_discard(data);
}

static void* pcm_take(size_t bytes, size_t* got)
{
    pthread_mutex_lock(&g_pool_lock);
    for (int i = 0; i < g_pool_n; i++)
        if (g_pool[i].bytes >= bytes && g_pool[i].bytes <= 2 * bytes + 4096) {
            void* p = g_pool[i].ptr; *got = g_pool[i].bytes;
            g_pool[i] = g_pool[--g_pool_n];
            pthread_mutex_unlock(&g_pool_lock);
            return p;
        }
    pthread_mutex_unlock(&g_pool_lock);
    void* p = NULL;
    pthread_mutex_lock(&g_prep_lock);
    if (!g_prep_started) {
        pthread_t t;
        if (pthread_create(&t, NULL, prep_main, NULL) == 0) { pthread_detach(t); g_prep_started = 1; }
    }
    for (int i = 0; i < g_ready_n && !p; i++)
        if (g_ready[i].bytes == bytes) { p = g_ready[i].ptr; g_ready[i] = g_ready[--g_ready_n]; }
    const int same_size = g_want_bytes == bytes;
    if (!same_size) {                              /* another batch size: what was made ahead for the old one goes */
        while (g_ready_n && g_trash_n < 64) g_trash[g_trash_n++] = g_ready[--g_ready_n].ptr;
        g_want_bytes = bytes;
    }
    if (g_prep_started) pthread_cond_signal(&g_prep_wake);
    /* none ready, but the helper is at one of this size: wait for that one rather than allocate beside it (two threads
     * page-locking 32 MiB at the same time take longer than one after the other: tools/micro/pinned_alloc.hip) */
    if (!p && g_prep_started && same_size) {
        struct timespec until;
        clock_gettime(CLOCK_REALTIME, &until);
        until.tv_nsec += 20 * 1000 * 1000;
        if (until.tv_nsec >= 1000000000L) { until.tv_sec++; until.tv_nsec -= 1000000000L; }
        while (!p && g_want_bytes == bytes) {
            for (int i = 0; i < g_ready_n && !p; i++)
                if (g_ready[i].bytes == bytes) { p = g_ready[i].ptr; g_ready[i] = g_ready[--g_ready_n]; }
            if (p) { pthread_cond_signal(&g_prep_wake); break; }
            if (pthread_cond_timedwait(&g_ready_wake, &g_prep_lock, &until) != 0) break;
        }
    }
    pthread_mutex_unlock(&g_prep_lock);
    *got = bytes;
    return p ? p : L.host_alloc(bytes);
}

/* pipelineDecode(pipeline, bytes:Uint8Array, frames:Uint32Array(2 * S * F) [offset, length]..., slots:Uint32Array(S), framesPerStream,
 *                results:Uint8Array(8 * S * F), channels [, ring, ringElems])
 *   -> { pcm: Float32Array|Int16Array(S * F * 1024 * channels) on page-locked memory, refused }
 * ring = 0 / absent: the PCM array's memory is the caller's for as long as any view of it lives (a buffer per call, recycled by
 * the garbage collector's finalizer).  ring = K: the pipeline's K buffers of ringElems elements, made at the first call, are
 * handed out in turn — what a call returns is overwritten by the K-th call after it. */
typedef struct {
    handle_box* pb;
    const uint8_t* db; size_t nb; const aacg_parse_frame* df; const uint32_t* ds; size_t ns; uint32_t F; aacg_parse_result* dr;
    void* pcm; size_t got, elems; int slot;
    napi_ref keep[4];                                  /* the four argument arrays: theirs is the memory the native call reads and writes */
    int n_keep;
    uint64_t ticket; int submitted;                    /* pipelineSubmit: aacg_pipeline_submit's ticket */
    int rc; uint32_t refused; char msg[1024];
} pipe_job;

/* argument checking and the PCM buffer: everything that needs the JavaScript thread before the native call */
static pipe_job* pipe_prepare(napi_env env, napi_callback_info info, int keep_args)
{
    size_t argc = 9; napi_value argv[9];
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok) return NULL;
    if (argc < 7) { napi_throw_error(env, NULL, "aacgpu: pipelineDecode takes at least 7 arguments"); return NULL; }
    uint32_t ring = 0, ring_elems = 0;
    if (argc >= 9) { napi_get_value_uint32(env, argv[7], &ring); napi_get_value_uint32(env, argv[8], &ring_elems); }
    handle_box* pb = box_of(env, argv[0], BOX_PIPELINE, "aacgpu: bad pipeline handle");
    if (!pb) return NULL;
    if (pb->n_jobs && !keep_args) { napi_throw_error(env, NULL, "aacgpu: submitted batches have not been collected yet (pipelineCollect)"); return NULL; }
    if (pb->n_jobs >= 3) { napi_throw_error(env, NULL, "aacgpu: pipelineSubmit: three batches are in flight already (pipelineCollect)"); return NULL; }
    napi_typedarray_type t; size_t nb, nf, ns, nr; void *db, *df, *ds, *dr; uint32_t F = 0, C = 0;
    napi_get_value_uint32(env, argv[4], &F);
    napi_get_value_uint32(env, argv[6], &C);
    if (!typed(env, argv[1], &t, &nb, &db) || t != napi_uint8_array || !typed(env, argv[2], &t, &nf, &df) || t != napi_uint32_array ||
        !typed(env, argv[3], &t, &ns, &ds) || t != napi_uint32_array || !F || nf != 2 * ns * F || C < 1 || C > AACG_MAX_CHANNELS ||
        !typed(env, argv[5], &t, &nr, &dr) || t != napi_uint8_array || nr != ns * F * sizeof(aacg_parse_result)) {
        napi_throw_type_error(env, NULL, "pipelineDecode(pipeline, Uint8Array bytes, Uint32Array frames (2 per frame), Uint32Array slots, framesPerStream, Uint8Array results (8 per frame), channels)");
        return NULL;
    }
    const size_t elems = ns * F * 1024u * C, bytes = elems * (pb->out_i16 ? 2u : 4u);
    size_t got = 0;
    void* pcm = NULL;
    int slot = -1;
    if (ring) {
        if (ring > PCM_RING_MAX || ring_elems < elems) { napi_throw_range_error(env, NULL, "aacgpu: pipelineDecode: ring of at most 16 buffers, each at least one batch long"); return NULL; }
        if (!pb->ring_n) {                                 /* the pipeline's ring: made once, kept alive by references until the pipeline goes */
            pb->ring_bytes = (size_t)ring_elems * (pb->out_i16 ? 2u : 4u);
            uint32_t made = 0;
            for (; made < ring; made++) {
                void* m = L.host_alloc(pb->ring_bytes);
                napi_value rab;
                if (!m) break;
                /* once the ArrayBuffer exists the memory is ITS (ring_finalize frees it when the buffer is collected); before, ours */
                if (napi_create_external_arraybuffer(env, m, pb->ring_bytes, ring_finalize, NULL, &rab) != napi_ok) { L.host_free(m); break; }
                if (napi_create_reference(env, rab, 1, &pb->ring_ab[made]) != napi_ok) break;
                pb->ring_ptr[made] = m;
            }
            if (made < ring) {                             /* all K or none: a shorter ring would overwrite frames earlier than the caller was told */
                for (uint32_t i = 0; i < made; i++) { napi_delete_reference(env, pb->ring_ab[i]); pb->ring_ab[i] = NULL; pb->ring_ptr[i] = NULL; }
                napi_throw_error(env, NULL, "aacgpu: out of page-locked memory (pcmRing)");
                return NULL;
            }
            pb->ring_n = (int)ring;
        }
        if (bytes > pb->ring_bytes) { napi_throw_range_error(env, NULL, "aacgpu: pipelineDecode: batch larger than the ring's buffers"); return NULL; }
        slot = (int)(pb->ring_next++ % (unsigned)pb->ring_n);
        pcm = pb->ring_ptr[slot]; got = pb->ring_bytes;
    } else {
        pcm = pcm_take(bytes, &got);
        if (!pcm) { napi_throw_error(env, NULL, "aacgpu: out of page-locked memory"); return NULL; }
    }
    pipe_job* j = (pipe_job*)calloc(1, sizeof *j);
    if (!j) { if (slot < 0 && !pool_put(pcm, got)) pcm_discard(pcm); napi_throw_error(env, NULL, "aacgpu: out of memory"); return NULL; }
    j->pb = pb; j->db = (const uint8_t*)db; j->nb = nb; j->df = (const aacg_parse_frame*)df; j->ds = (const uint32_t*)ds; j->ns = ns; j->F = F;
    j->dr = (aacg_parse_result*)dr; j->pcm = pcm; j->got = got; j->elems = elems; j->slot = slot;
    if (keep_args) {
        const int which[4] = {1, 2, 3, 5};
        for (int k = 0; k < 4; k++)
            if (napi_create_reference(env, argv[which[k]], 1, &j->keep[j->n_keep]) == napi_ok) j->n_keep++;
    }
    return j;
}

/* the synchronous call (pipelineDecode), or the enqueue half of the asynchronous pair (pipelineSubmit: aacg_pipeline_submit
 * stages the bytes and returns when the batch is on its lane — no thread of the addon's is involved, round 5's was) */
static void pipe_run(pipe_job* j, int submit_only)
{
    pthread_mutex_lock(&j->pb->lock);
    aacg_pipeline* p = (aacg_pipeline*)j->pb->ptr;
    if (submit_only) {
        j->rc = L.pipeline_submit(p, j->db, j->nb, j->df, j->ds, (uint32_t)j->ns, j->F, j->pcm, j->dr, &j->refused, &j->ticket);
        j->submitted = j->rc == 0;
    } else j->rc = L.pipeline_decode(p, j->db, j->nb, j->df, j->ds, (uint32_t)j->ns, j->F, j->pcm, j->dr, &j->refused);
    if (j->rc) snprintf(j->msg, sizeof j->msg, "aacgpu: %s failed (%d): %.800s", submit_only ? "aacg_pipeline_submit" : "aacg_pipeline_decode", j->rc, L.pipeline_last_error(p));
    pthread_mutex_unlock(&j->pb->lock);
}
static void pipe_wait(pipe_job* j)
{
    if (!j->submitted) return;
    pthread_mutex_lock(&j->pb->lock);
    aacg_pipeline* p = (aacg_pipeline*)j->pb->ptr;
    j->rc = L.pipeline_collect(p, j->ticket);
    if (j->rc) snprintf(j->msg, sizeof j->msg, "aacgpu: aacg_pipeline_collect failed (%d): %.800s", j->rc, L.pipeline_last_error(p));
    j->submitted = 0;
    pthread_mutex_unlock(&j->pb->lock);
}

/* back on the JavaScript thread: { pcm, refused } or the error; the job is freed */
static napi_value pipe_finish(napi_env env, pipe_job* j)
{
    napi_value out = NULL, v, ab;
    handle_box* pb = j->pb;
    for (int k = 0; k < j->n_keep; k++) napi_delete_reference(env, j->keep[k]);
    if (j->rc) {
        if (j->slot < 0 && !pool_put(j->pcm, j->got)) pcm_discard(j->pcm);
        napi_throw_error(env, NULL, j->msg);
        free(j);
        return NULL;
    }
    const size_t elems = j->elems, got = j->got; void* pcm = j->pcm; const int slot = j->slot; const uint32_t refused = j->refused;
    free(j);
    if (slot >= 0) CHECK(env, napi_get_reference_value(env, pb->ring_ab[slot], &ab));
    else
    /* (the engine is not told about the buffer's size with napi_adjust_external_memory: it answers 33 MB of external memory
     * per batch with a full collection per batch; the views are short-lived objects, ordinary collections find the buffers) */
    if (napi_create_external_arraybuffer(env, pcm, got, pcm_finalize, (void*)got, &ab) != napi_ok) { L.host_free(pcm); napi_throw_error(env, NULL, "aacgpu: napi_create_external_arraybuffer failed"); return NULL; }
    CHECK(env, napi_create_typedarray(env, pb->out_i16 ? napi_int16_array : napi_float32_array, elems, ab, 0, &v));
    CHECK(env, napi_create_object(env, &out));
    CHECK(env, napi_set_named_property(env, out, "pcm", v));
    CHECK(env, napi_create_uint32(env, refused, &v));
    CHECK(env, napi_set_named_property(env, out, "refused", v));
    return out;
}

static void pipe_abandon(napi_env env, void* job)
{
    pipe_job* j = (pipe_job*)job;
    pipe_wait(j);                                       /* the device writes into j->pcm and the results array until then */
    for (int k = 0; k < j->n_keep; k++) napi_delete_reference(env, j->keep[k]);
    if (j->slot < 0) pcm_discard(j->pcm);
    free(j);
}

static napi_value js_pipeline_decode(napi_env env, napi_callback_info info)
{
    pipe_job* j = pipe_prepare(env, info, 0);
    if (!j) return NULL;
    pipe_run(j, 0);
    return pipe_finish(env, j);
}

/* pipelineSubmit(same arguments as pipelineDecode): the batch is staged and enqueued (aacg_pipeline_submit) and the call returns
 * while the device works — JavaScript slices the batch before, say.  The bytes / frames / slots arrays may be reused at once; the
 * results array belongs to the batch until pipelineCollect(pipeline) -> { pcm, refused } has returned the OLDEST submitted batch
 * (it waits for it, bounded).  Up to three batches in flight per pipeline. */
static napi_value js_pipeline_submit(napi_env env, napi_callback_info info)
{
    pipe_job* j = pipe_prepare(env, info, 1);
    if (!j) return NULL;
    pipe_run(j, 1);
    if (j->rc) return pipe_finish(env, j);              /* refused at once: thrown here, nothing to collect */
    j->pb->jobs[j->pb->n_jobs++] = j;
    return NULL;
}

static napi_value js_pipeline_collect(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1];
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_box* pb = box_of(env, argv[0], BOX_PIPELINE, "aacgpu: bad pipeline handle");
    if (!pb) return NULL;
    if (!pb->n_jobs) { napi_throw_error(env, NULL, "aacgpu: pipelineCollect: nothing has been submitted"); return NULL; }
    pipe_job* j = (pipe_job*)pb->jobs[0];
    for (int i = 1; i < pb->n_jobs; i++) pb->jobs[i - 1] = pb->jobs[i];
    pb->n_jobs--;
    pipe_wait(j);
    return pipe_finish(env, j);
}

static napi_value js_pipeline_reset_stream(napi_env env, napi_callback_info info)
{
    size_t argc = 2; napi_value argv[2]; uint32_t slot = 0;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    handle_box* pb = box_of(env, argv[0], BOX_PIPELINE, "aacgpu: bad pipeline handle");
    if (!pb) return NULL;
    napi_get_value_uint32(env, argv[1], &slot);
    pthread_mutex_lock(&pb->lock);
    int rc = L.pipeline_reset_stream((aacg_pipeline*)pb->ptr, slot);
    pthread_mutex_unlock(&pb->lock);
    if (rc) napi_throw_error(env, NULL, "aacgpu: aacg_pipeline_reset_stream failed");
    return NULL;
}

/* parseStatusString(code) -> the reference's message for a per-frame status */
static napi_value js_parse_status_string(napi_env env, napi_callback_info info)
{
    size_t argc = 1; napi_value argv[1], out; int32_t code = 0;
    CHECK(env, napi_get_cb_info(env, info, &argc, argv, NULL, NULL));
    if (!L.dl) { napi_throw_error(env, NULL, "aacgpu: call load(path) first"); return NULL; }
    napi_get_value_int32(env, argv[0], &code);
    CHECK(env, napi_create_string_utf8(env, L.parse_status_string(code), NAPI_AUTO_LENGTH, &out));
    return out;
}

static napi_value init(napi_env env, napi_value exports)
{
    napi_property_descriptor props[] = {
        {"load", NULL, js_load, NULL, NULL, NULL, napi_default, NULL},
        {"create", NULL, js_create, NULL, NULL, NULL, napi_default, NULL},
        {"decodeBatch", NULL, js_decode_batch, NULL, NULL, NULL, napi_default, NULL},
        {"decodeBatchAsync", NULL, js_decode_batch_async, NULL, NULL, NULL, napi_default, NULL},
        {"resetStream", NULL, js_reset_stream, NULL, NULL, NULL, napi_default, NULL},
        {"getOverlap", NULL, js_get_overlap, NULL, NULL, NULL, napi_default, NULL},
        {"setOverlap", NULL, js_set_overlap, NULL, NULL, NULL, napi_default, NULL},
        {"parserCreate", NULL, js_parser_create, NULL, NULL, NULL, napi_default, NULL},
        {"parseBatch", NULL, js_parse_batch, NULL, NULL, NULL, napi_default, NULL},
        {"parseStatusString", NULL, js_parse_status_string, NULL, NULL, NULL, napi_default, NULL},
        {"pipelineCreate", NULL, js_pipeline_create, NULL, NULL, NULL, napi_default, NULL},
        {"pipelineDecode", NULL, js_pipeline_decode, NULL, NULL, NULL, napi_default, NULL},
        {"pipelineSubmit", NULL, js_pipeline_submit, NULL, NULL, NULL, napi_default, NULL},
        {"pipelineCollect", NULL, js_pipeline_collect, NULL, NULL, NULL, napi_default, NULL},
        {"pipelineResetStream", NULL, js_pipeline_reset_stream, NULL, NULL, NULL, napi_default, NULL},
    };
    napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
    return exports;
}

NAPI_MODULE(NODE_GYP_MODULE_NAME, init)
