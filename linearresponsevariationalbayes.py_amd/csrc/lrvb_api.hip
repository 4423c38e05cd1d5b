// lrvb_api.hip -- the C ABI declared in include/lrvb_hip.h (context, orchestration, host<->device
// staging).  Every kernel lives in a k_*.hip file; the ones this file launches itself are declared in k_kernels.h.
#include "lrvb_internal.h"
#include "k_kernels.h"
#include <vector>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <new>
#include <algorithm>

static thread_local char g_err[1024] = "";

void lrvb_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* lrvb_last_error(void) { return g_err; }
extern "C" int lrvb_version(void) { return LRVB_ABI_VERSION; }
extern "C" int lrvb_device_count(int* out) {
    if (!out) LRVB_FAIL(LRVB_ERR_INVALID, "null out");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *out = n;
    return LRVB_OK;
}

int buf_reserve(lrvb_ctx* c, DevBuf& b, size_t n) {
    if (n == 0) n = 1;
    if (b.p != nullptr && b.n >= n && b.owned) return LRVB_OK;
    if (b.p != nullptr && !b.owned && b.n >= n) return LRVB_OK;
    if (b.p != nullptr && b.owned) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(b.p));
    }
    b.p = nullptr; b.n = 0; b.owned = true;
    HIP_TRY(hipMalloc((void**)&b.p, n * sizeof(double)));
    b.n = n;
    ++c->buf_epoch;                                      // captured graphs hold the old address
    return LRVB_OK;
}
int reserve_obs_vec(lrvb_ctx* c, DevBuf& b) {
    const size_t need = (size_t)c->N + 64;
    if (b.p != nullptr && b.owned && b.n >= need) return LRVB_OK;
    LRVB_TRY(buf_reserve(c, b, need));
    HIP_TRY(hipMemsetAsync(b.p + c->N, 0, 64 * sizeof(double), c->stream));
    return LRVB_OK;
}
void buf_free(DevBuf& b) {
    if (b.p && b.owned) (void)hipFree(b.p);
    b.p = nullptr; b.n = 0; b.owned = true;
}

static void comm_release(lrvb_ctx* c);      // destroys the context's RCCL communicator, if any (defined with the RCCL loader)

static int ctx_bind(lrvb_ctx* c) {
    if (!c) LRVB_FAIL(LRVB_ERR_INVALID, "null context");
    c->hvp_pt_valid = false;
    HIP_TRY(hipSetDevice(c->device));
    return LRVB_OK;
}

static int pinned_reserve(lrvb_ctx* c, size_t n) {
    if (c->host_pinned_n >= n) return LRVB_OK;
    if (c->host_pinned) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipHostFree(c->host_pinned)); c->host_pinned = nullptr; c->host_pinned_n = 0; }
    HIP_TRY(hipHostMalloc((void**)&c->host_pinned, n * sizeof(double), hipHostMallocDefault));
    c->host_pinned_n = n;
    return LRVB_OK;
}

// the context's side stream (non-blocking: ordered against nothing but its own events)
static int ensure_aux(lrvb_ctx* c) {
    if (!c->aux_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) HIP_TRY(hipEventCreateWithFlags(&c->aux_ev[k], hipEventDisableTiming));
    }
    return LRVB_OK;
}
// host -> device on the side stream, complete on return: the copy runs BESIDE whatever is queued on the context's stream
// (the caller guarantees that nothing queued there touches dst)
static int h2d_beside(lrvb_ctx* c, double* dst, const double* src, size_t n) {
    if (n == 0) return LRVB_OK;
    LRVB_TRY(ensure_aux(c));
    HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, c->aux_stream));
    HIP_TRY(hipStreamSynchronize(c->aux_stream));
    return LRVB_OK;
}
static int h2d(lrvb_ctx* c, double* dst, const double* src, size_t n) {
    if (n == 0) return LRVB_OK;
    if (n <= lrvb_ctx::UP_SLOT_DOUBLES) {
        // through a pinned slot: the caller's (pageable) buffer is consumed by the memcpy, the device copy is stream-ordered and
        // nobody waits for it -- a synchronising upload cost 10-20 us of host time per call, a dozen times per step in the
        // small configurations.  A slot is reused only after the event behind its last copy has completed.
        if (!c->up_ring) {
            HIP_TRY(hipHostMalloc((void**)&c->up_ring, lrvb_ctx::UP_SLOTS * lrvb_ctx::UP_SLOT_DOUBLES * sizeof(double), hipHostMallocDefault));
            for (int k = 0; k < lrvb_ctx::UP_SLOTS; ++k) HIP_TRY(hipEventCreateWithFlags(&c->up_ev[k], hipEventDisableTiming));
            HIP_TRY(hipHostGetDevicePointer((void**)&c->up_ring_dev, c->up_ring, 0));
        }
        const int k = c->up_next;
        c->up_next = (k + 1) % lrvb_ctx::UP_SLOTS;
        HIP_TRY(hipEventSynchronize(c->up_ev[k]));            // returns at once for an event that was never recorded
        double* slot = c->up_ring + (size_t)k * lrvb_ctx::UP_SLOT_DOUBLES;
        memcpy(slot, src, n * sizeof(double));
        hipLaunchKernelGGL(upload_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, dst,
                           (const double*)(c->up_ring_dev + (size_t)k * lrvb_ctx::UP_SLOT_DOUBLES), (i64)n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->up_ev[k], c->stream));
        return LRVB_OK;
    }
    HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));      // src is pageable caller memory
    return LRVB_OK;
}
static int d2h(lrvb_ctx* c, double* dst, const double* src, size_t n) {
    if (n == 0) return LRVB_OK;
    if (n <= 4096) {          // scalars of the iterative solvers: through the context's pinned page (no pageable staging in the runtime)
        LRVB_TRY(pinned_reserve(c, 4096));
        HIP_TRY(hipMemcpyAsync(c->host_pinned, src, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        memcpy(dst, c->host_pinned, n * sizeof(double));
        return LRVB_OK;
    }
    HIP_TRY(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LRVB_OK;
}

// ---- context -------------------------------------------------------------------------
extern "C" int lrvb_ctx_create(lrvb_ctx** out, int device_id, const lrvb_model_desc* m) {
    if (!out || !m) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    *out = nullptr;
    if (m->n_blocks <= 0 || !m->blocks) LRVB_FAIL(LRVB_ERR_INVALID, "model needs at least one parameter block");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device_id < 0 || device_id >= ndev) LRVB_FAIL(LRVB_ERR_INVALID, "device %d out of range (%d devices)", device_id, ndev);
    lrvb_ctx* c = new (std::nothrow) lrvb_ctx();
    if (!c) LRVB_FAIL(LRVB_ERR_HIP, "out of host memory");
    c->device = device_id;
    i64 D = 0, V = 0;
    for (int i = 0; i < m->n_blocks; ++i) {
        lrvb_block_desc b = m->blocks[i];
        if (b.free_off != D || b.vec_off != V) { delete c; LRVB_FAIL(LRVB_ERR_INVALID, "block %d: offsets must be the running sums of the preceding sizes", i); }
        if (b.kind == LRVB_BLOCK_BOX) {
            if (b.free_size != b.vec_size || b.free_size < 0) { delete c; LRVB_FAIL(LRVB_ERR_INVALID, "block %d: box sizes", i); }
            if (!(b.lb < b.ub)) { delete c; LRVB_FAIL(LRVB_ERR_INVALID, "block %d: upper bound must strictly exceed lower bound", i); }
        } else if (b.kind == LRVB_BLOCK_PSD) {
            const i64 k = b.dim0;
            if (k <= 0 || b.free_size != k * (k + 1) / 2 || b.vec_size != b.free_size || b.lb < 0.0) { delete c; LRVB_FAIL(LRVB_ERR_INVALID, "block %d: psd sizes", i); }
            c->all_box = false;
        } else if (b.kind == LRVB_BLOCK_SIMPLEX) {
            if (b.dim0 <= 0 || b.dim1 < 2 || b.free_size != b.dim0 * (b.dim1 - 1) || b.vec_size != b.dim0 * b.dim1) { delete c; LRVB_FAIL(LRVB_ERR_INVALID, "block %d: simplex sizes", i); }
            c->all_box = false;
        } else { delete c; LRVB_FAIL(LRVB_ERR_INVALID, "block %d: unknown kind %d", i, b.kind); }
        D += b.free_size; V += b.vec_size;
        c->blocks.push_back(b);
    }
    c->D = D; c->V = V;
    c->loss = m->loss;
    if (m->loss != LRVB_LOSS_NONE) {
        if (m->loss < LRVB_LOSS_GAUSSIAN || m->loss > LRVB_LOSS_DATA_ONLY) { delete c; LRVB_FAIL(LRVB_ERR_INVALID, "unknown loss %d", m->loss); }
        if (m->n_obs <= 0 || m->n_cols <= 0) { delete c; LRVB_FAIL(LRVB_ERR_INVALID, "data term needs n_obs > 0 and n_cols > 0"); }
        if (m->loss != LRVB_LOSS_DATA_ONLY && (m->glm_off < 0 || m->glm_off + m->n_cols > V)) { delete c; LRVB_FAIL(LRVB_ERR_SIZE, "coefficient slice [%lld, %lld) exceeds vector size %lld", (long long)m->glm_off, (long long)(m->glm_off + m->n_cols), (long long)V); }
        c->N = m->n_obs; c->P = m->n_cols; c->glm_off = m->glm_off; c->lik_info = m->lik_info;
    }
    c->quad_kind = m->quad_kind;
    if (c->quad_kind < LRVB_QUAD_NONE || c->quad_kind > LRVB_QUAD_DENSE) { delete c; LRVB_FAIL(LRVB_ERR_INVALID, "unknown quad_kind"); }
    if (c->loss == LRVB_LOSS_NONE && c->quad_kind == LRVB_QUAD_NONE) { delete c; LRVB_FAIL(LRVB_ERR_INVALID, "model has neither a data term nor a quadratic term"); }
    c->data_only = (c->loss == LRVB_LOSS_DATA_ONLY);

    hipError_t e = hipSetDevice(device_id);
    // A BLOCKING stream (hipStreamDefault): it orders itself against the legacy default stream in both directions, so a
    // caller that produces operands on the default stream (torch's current stream unless told otherwise) and hands their
    // device pointers to a `_dev` entry point needs no further synchronisation (include/lrvb_hip.h, "stream ordering").
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream, hipStreamDefault);
    if (e != hipSuccess) { lrvb_set_error("context init: %s", hipGetErrorString(e)); delete c; return LRVB_ERR_HIP; }

    int st = upload_boxmap(c);
    if (st == LRVB_OK) st = upload_jtmap(c);
    auto need = [&](DevBuf& b, size_t n) { if (st == LRVB_OK) st = buf_reserve(c, b, n); };
    need(c->theta, (size_t)(V > D ? V : D)); need(c->eta, (size_t)V); need(c->j1, (size_t)D); need(c->j2, (size_t)D);
    need(c->g_eta, (size_t)V); need(c->g_free, (size_t)D);
    need(c->vtmp, (size_t)(V > D ? V : D)); need(c->vtmp2, (size_t)(V > D ? V : D)); need(c->vtmp3, (size_t)(V > D ? V : D));
    need(c->scal, 16);
    if (c->loss != LRVB_LOSS_NONE) {
        need(c->w, (size_t)c->N);
        need(c->lp, (size_t)c->N);
        if (st == LRVB_OK) st = reserve_obs_vec(c, c->cw);
        const size_t tiles = (size_t)wsyrk_num_tiles(c->P) * WS_TILE * WS_TILE;
        need(c->stats, 1 + (size_t)c->P + tiles);
    } else {
        need(c->stats, 2);
    }
    if (c->quad_kind != LRVB_QUAD_NONE) {
        need(c->quadA, c->quad_kind == LRVB_QUAD_DIAG ? (size_t)V : (size_t)V * (size_t)V);
        need(c->quadM, (size_t)V); need(c->quadB, (size_t)V);
    }
    if (st == LRVB_OK) {
        if (c->loss != LRVB_LOSS_NONE) {
            // w = 1
            std::vector<double> ones((size_t)c->N, 1.0);
            st = h2d(c, c->w.p, ones.data(), (size_t)c->N);
        }
    }
    if (st == LRVB_OK && c->quad_kind != LRVB_QUAD_NONE) {
        if (hipMemsetAsync(c->quadA.p, 0, c->quadA.n * sizeof(double), c->stream) != hipSuccess ||
            hipMemsetAsync(c->quadM.p, 0, (size_t)V * sizeof(double), c->stream) != hipSuccess ||
            hipMemsetAsync(c->quadB.p, 0, (size_t)V * sizeof(double), c->stream) != hipSuccess) {
            lrvb_set_error("memset failed"); st = LRVB_ERR_HIP;
        }
    }
    if (st != LRVB_OK) { lrvb_ctx_destroy(c); return st; }
    *out = c;
    return LRVB_OK;
}

extern "C" int lrvb_ctx_destroy(lrvb_ctx* c) {
    if (!c) return LRVB_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    comm_release(c);
    DevBuf* all[] = { &c->X, &c->y, &c->w, &c->quadA, &c->quadM, &c->quadB, &c->theta, &c->eta, &c->j1, &c->j2,
                      &c->vtmp, &c->vtmp2, &c->vtmp3, &c->g_eta, &c->g_free, &c->lp, &c->cw, &c->zbuf,
                      &c->part_vec, &c->part_val, &c->stats, &c->tile_part, &c->Heta, &c->Hfree, &c->Jdense,
                      &c->Tdense, &c->work1, &c->chol, &c->cholW, &c->hprog, &c->cgH, &c->groups, &c->mx_theta, &c->mx_lam, &c->mx_A, &c->mx_U, &c->mx_g, &c->mx_Xk, &c->mx_R, &c->cgT, &c->ones, &c->cgm[0], &c->cgm[1], &c->cgm[2], &c->cgm[3], &c->cgm[4], &c->cgm[5], &c->cgm[6], &c->cgm[7], &c->cgm[8], &c->rhs, &c->cgx, &c->cgr, &c->cgp, &c->cgq, &c->cgz, &c->scal, &c->opt, &c->dkw, &c->cyv, &c->rvec, &c->red_scratch, &c->gstats, &c->Zs, &c->ws, &c->bpart, &c->gpad, &c->boxmap, &c->jtmap, &c->qg_Mt, &c->qg_T1, &c->qg_Av, &c->Hres, &c->hres_theta, &c->qstats };
    for (DevBuf* b : all) buf_free(*b);
    if (c->host_pinned) (void)hipHostFree(c->host_pinned);
    if (c->up_ring) { for (int k = 0; k < lrvb_ctx::UP_SLOTS; ++k) if (c->up_ev[k]) (void)hipEventDestroy(c->up_ev[k]); (void)hipHostFree(c->up_ring); }
    for (int k = 0; k < PROF_POOLS; ++k) for (hipEvent_t e : c->ev_pool[k]) (void)hipEventDestroy(e);
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    for (int k = 0; k < 2; ++k) if (c->aux_ev[k]) (void)hipEventDestroy(c->aux_ev[k]);
    if (c->mv_graph.exec) (void)hipGraphExecDestroy(c->mv_graph.exec);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->stream && c->stream_owned) (void)hipStreamDestroy(c->stream);
    delete c;
    return LRVB_OK;
}

extern "C" int lrvb_ctx_sync(lrvb_ctx* c) {
    LRVB_TRY(ctx_bind(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LRVB_OK;
}

extern "C" int lrvb_ctx_set_stream(lrvb_ctx* c, void* hip_stream, int use_caller_stream) {
    LRVB_TRY(ctx_bind(c));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->stream && c->stream_owned) HIP_TRY(hipStreamDestroy(c->stream));
    if (!use_caller_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamDefault));
        c->stream_owned = true;
    } else {
        c->stream = reinterpret_cast<hipStream_t>(hip_stream);
        c->stream_owned = false;
    }
    return LRVB_OK;
}

// Event hand-off between the context's stream and another stream of the same device (no host synchronisation).
static int stream_handoff(lrvb_ctx* c, hipStream_t from, hipStream_t to) {
    if (from == to) return LRVB_OK;
    if (!c->ev_order) HIP_TRY(hipEventCreateWithFlags(&c->ev_order, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(c->ev_order, from));
    HIP_TRY(hipStreamWaitEvent(to, c->ev_order, 0));
    return LRVB_OK;
}
extern "C" int lrvb_ctx_wait_stream(lrvb_ctx* c, void* hip_stream) {
    LRVB_TRY(ctx_bind(c));
    return stream_handoff(c, reinterpret_cast<hipStream_t>(hip_stream), c->stream);
}
extern "C" int lrvb_stream_wait_ctx(lrvb_ctx* c, void* hip_stream) {
    LRVB_TRY(ctx_bind(c));
    return stream_handoff(c, c->stream, reinterpret_cast<hipStream_t>(hip_stream));
}

extern "C" int lrvb_ctx_sizes(lrvb_ctx* c, int64_t* D, int64_t* V, int64_t* n_obs) {
    if (!c) LRVB_FAIL(LRVB_ERR_INVALID, "null context");
    if (D) *D = c->D;
    if (V) *V = c->V;
    if (n_obs) *n_obs = c->N;
    return LRVB_OK;
}

static int slot_shape_check(lrvb_ctx* c, int slot, i64 rows, i64 cols, DevBuf** buf, size_t* n) {
    switch (slot) {
    case LRVB_SLOT_X:
        if (c->loss == LRVB_LOSS_NONE) LRVB_FAIL(LRVB_ERR_STATE, "model has no data term");
        if (rows != c->N || cols != c->P) LRVB_FAIL(LRVB_ERR_SIZE, "X must be %lld x %lld (got %lld x %lld)", (long long)c->N, (long long)c->P, (long long)rows, (long long)cols);
        *buf = &c->X; *n = (size_t)rows * (size_t)cols; return LRVB_OK;
    case LRVB_SLOT_Y:
        if (c->loss == LRVB_LOSS_NONE) LRVB_FAIL(LRVB_ERR_STATE, "model has no data term");
        if (rows * cols != c->N) LRVB_FAIL(LRVB_ERR_SIZE, "y must have %lld entries (got %lld)", (long long)c->N, (long long)(rows * cols));
        *buf = &c->y; *n = (size_t)c->N; return LRVB_OK;
    case LRVB_SLOT_QUAD_A:
        if (c->quad_kind == LRVB_QUAD_NONE) LRVB_FAIL(LRVB_ERR_STATE, "model has no quadratic term");
        if (c->quad_kind == LRVB_QUAD_DIAG) { if (rows * cols != c->V) LRVB_FAIL(LRVB_ERR_SIZE, "diagonal A must have %lld entries", (long long)c->V); *n = (size_t)c->V; }
        else { if (rows != c->V || cols != c->V) LRVB_FAIL(LRVB_ERR_SIZE, "A must be %lld x %lld", (long long)c->V, (long long)c->V); *n = (size_t)c->V * (size_t)c->V; }
        *buf = &c->quadA; return LRVB_OK;
    case LRVB_SLOT_QUAD_M:
    case LRVB_SLOT_QUAD_B:
        if (c->quad_kind == LRVB_QUAD_NONE) LRVB_FAIL(LRVB_ERR_STATE, "model has no quadratic term");
        if (rows * cols != c->V) LRVB_FAIL(LRVB_ERR_SIZE, "vector must have %lld entries", (long long)c->V);
        *buf = (slot == LRVB_SLOT_QUAD_M) ? &c->quadM : &c->quadB; *n = (size_t)c->V; return LRVB_OK;
    default:
        LRVB_FAIL(LRVB_ERR_INVALID, "unknown data slot %d", slot);
    }
}

extern "C" int lrvb_set_data(lrvb_ctx* c, int slot, const double* host, int64_t rows, int64_t cols) {
    LRVB_TRY(ctx_bind(c));
    if (!host) LRVB_FAIL(LRVB_ERR_INVALID, "null data");
    DevBuf* b = nullptr; size_t n = 0;
    LRVB_TRY(slot_shape_check(c, slot, rows, cols, &b, &n));
    if (!b->owned) { b->p = nullptr; b->n = 0; b->owned = true; }
    LRVB_TRY(buf_reserve(c, *b, n));
    LRVB_TRY(h2d(c, b->p, host, n));
    if (slot == LRVB_SLOT_X) { c->have_X = true; c->x2_ready = false; c->gstats_valid = false; c->zs_valid = false; }
    if (slot == LRVB_SLOT_Y) c->have_y = true;
    c->hres_valid = false;
    return LRVB_OK;
}

extern "C" int lrvb_set_data_dev(lrvb_ctx* c, int slot, const double* data_dev, int64_t rows, int64_t cols) {
    LRVB_TRY(ctx_bind(c));
    if (!data_dev) LRVB_FAIL(LRVB_ERR_INVALID, "null data");
    DevBuf* b = nullptr; size_t n = 0;
    LRVB_TRY(slot_shape_check(c, slot, rows, cols, &b, &n));
    if (b->p && b->owned) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(b->p)); }
    b->p = const_cast<double*>(data_dev); b->n = n; b->owned = false; ++c->buf_epoch;
    if (slot == LRVB_SLOT_X) { c->have_X = true; c->x2_ready = false; c->gstats_valid = false; c->zs_valid = false; }
    if (slot == LRVB_SLOT_Y) c->have_y = true;
    c->hres_valid = false;
    return LRVB_OK;
}

extern "C" int lrvb_set_weights(lrvb_ctx* c, const double* w, int64_t n) {
    LRVB_TRY(ctx_bind(c));
    if (c->loss == LRVB_LOSS_NONE) LRVB_FAIL(LRVB_ERR_STATE, "model has no data term");
    if (!w || n != c->N) LRVB_FAIL(LRVB_ERR_SIZE, "weights must have %lld entries (got %lld)", (long long)c->N, (long long)n);
    if (!c->w.owned) { c->w.p = nullptr; c->w.n = 0; c->w.owned = true; }
    c->gstats_valid = false; c->ws_valid = false; c->hres_valid = false;
    LRVB_TRY(buf_reserve(c, c->w, (size_t)n));
    return h2d(c, c->w.p, w, (size_t)n);
}

extern "C" int lrvb_set_weights_dev(lrvb_ctx* c, const double* w_dev, int64_t n) {
    LRVB_TRY(ctx_bind(c));
    if (c->loss == LRVB_LOSS_NONE) LRVB_FAIL(LRVB_ERR_STATE, "model has no data term");
    if (!w_dev || n != c->N) LRVB_FAIL(LRVB_ERR_SIZE, "weights must have %lld entries (got %lld)", (long long)c->N, (long long)n);
    if (c->w.p && c->w.owned) { HIP_TRY(hipStreamSynchronize(c->stream)); HIP_TRY(hipFree(c->w.p)); }
    c->w.p = const_cast<double*>(w_dev); c->w.n = (size_t)n; c->w.owned = false; ++c->buf_epoch;
    c->gstats_valid = false; c->ws_valid = false; c->hres_valid = false;
    return LRVB_OK;
}

extern "C" int lrvb_set_quad_scale(lrvb_ctx* c, double scale) {
    if (c) c->hvp_pt_valid = false;
    if (!c) LRVB_FAIL(LRVB_ERR_INVALID, "null context");
    if (scale != c->quad_scale) c->hres_valid = false;
    c->quad_scale = scale;
    return LRVB_OK;
}

extern "C" int lrvb_set_lik_info(lrvb_ctx* c, double lik_info) {
    if (!c) LRVB_FAIL(LRVB_ERR_INVALID, "null context");
    c->hvp_pt_valid = false;
    if (c->loss != LRVB_LOSS_GAUSSIAN) LRVB_FAIL(LRVB_ERR_STATE, "lik_info is the precision of the Gaussian loss");
    if (!(lik_info > 0.0) || !std::isfinite(lik_info)) LRVB_FAIL(LRVB_ERR_INVALID, "lik_info must be positive and finite");
    c->lik_info = lik_info;
    c->hres_valid = false;
    return LRVB_OK;
}

extern "C" int lrvb_set_tuning(lrvb_ctx* c, int n_splits, int reserved) {
    if (!c) LRVB_FAIL(LRVB_ERR_INVALID, "null context");
    if (n_splits < 0 || n_splits > 1024) LRVB_FAIL(LRVB_ERR_INVALID, "n_splits out of range");
    if (reserved & ~15) LRVB_FAIL(LRVB_ERR_INVALID, "reserved = %d: only bits 0-3 are defined (include/lrvb_hip.h)", reserved);
    c->hvp_pt_valid = false;
    c->no_resident = (reserved & 8) != 0;      // (the resident matrix itself stays: it is the Hessian whichever kernels would form it)
    c->n_splits_user = n_splits;
    c->force_generic_wsyrk = (reserved & 1) != 0;
    c->force_dense_rows = (reserved & 2) ? 1 : 0;
    c->hm_four_waves = (reserved & 4) != 0;
    return LRVB_OK;
}

static int data_ready(lrvb_ctx* c) {
    if (c->data_only)
        LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "this context only holds data (LRVB_LOSS_DATA_ONLY): use lrvb_weighted_gram / lrvb_obs_quadform and the packing entry points");
    if (c->loss != LRVB_LOSS_NONE && !(c->have_X && c->have_y))
        LRVB_FAIL(LRVB_ERR_STATE, "observations not set: call lrvb_set_data for LRVB_SLOT_X and LRVB_SLOT_Y first");
    return LRVB_OK;
}

// ---- small elementwise kernels used only by the orchestration -------------------------
static inline unsigned nb256(i64 n) { return (unsigned)((n + 255) / 256); }
#define EW(kernel, n, ...) do { if ((n) > 0) { hipLaunchKernelGGL(kernel, dim3(nb256(n)), dim3(256), 0, c->stream, n, __VA_ARGS__); HIP_TRY(hipGetLastError()); } } while (0)

// ---- evaluation state -----------------------------------------------------------------
// eta (and j1/j2 for box blocks) from theta; in vector mode eta is the input itself.
static int set_point(lrvb_ctx* c, const double* point_dev, bool is_free) {
    if (is_free) {
        LRVB_TRY(launch_constrain(c, point_dev, c->eta.p, c->j1.p, c->j2.p));
    } else {
        HIP_TRY(hipMemcpyAsync(c->eta.p, point_dev, (size_t)c->V * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    return LRVB_OK;
}

// Sum of a device buffer of observation sums over the ranks of the job (no-op in a single process).
static int obs_reduce(lrvb_ctx* c, double* buf_dev, i64 n) {
    if (!c->reduce_fn || n <= 0) return LRVB_OK;
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_REDUCE));          // the exchange, timed apart from the kernels around it
    const int st = c->reduce_fn(c->reduce_user, buf_dev, (int64_t)n, (void*)c->stream);
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_REDUCE));
    if (st != 0) LRVB_FAIL(LRVB_ERR_STATE, "the reduce hook failed with status %d", st);
    return LRVB_OK;
}
extern "C" int lrvb_set_reduce_hook(lrvb_ctx* c, lrvb_reduce_fn fn, void* user) {
    if (!c) LRVB_FAIL(LRVB_ERR_INVALID, "null context");
    c->hvp_pt_valid = false; c->hres_valid = false;
    c->reduce_fn = fn; c->reduce_user = fn ? user : nullptr;
    return LRVB_OK;
}

// ---- in-library collective: RCCL, loaded at run time --------------------------------------------------------------
// One process per GPU; the ranks of the job exchange a 128-byte id out of band (any channel: the Python layer uses
// torch.distributed's store) and each creates its rank of ONE RCCL communicator here.  The communicator then serves
// as the native sum-over-ranks hook (ncclAllReduce on the context's stream, in place) and behind
// lrvb_allreduce_hessian.  librccl is dlopen'ed by soname -- if the process has already mapped a copy (torch's), that
// copy is used -- so liblrvb_hip.so keeps its single link dependency (libamdhip64).
#include <dlfcn.h>
namespace {
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, lrvb_comm_id, int) = nullptr;      // ncclUniqueId is a 128-byte struct passed by value
    int (*CommDestroy)(void*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int rccl_load() {
    if (g_rccl.lib) return LRVB_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "librccl.so.1 cannot be loaded: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(void*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void**, int, lrvb_comm_id, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(void*))dlsym(h, "ncclCommDestroy");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllReduce)
        LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "librccl lacks an expected entry point");
    g_rccl.lib = h;
    return LRVB_OK;
}
#define RCCL_TRY(expr) do { int r_ = (expr); if (r_ != 0) { \
    lrvb_set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error"); return LRVB_ERR_HIP; } } while (0)
int native_reduce(void* user, double* buf, int64_t n, void* stream) {
    lrvb_ctx* c = static_cast<lrvb_ctx*>(user);
    // ncclSum = 0, ncclDouble = 8 (rccl.h)
    return g_rccl.AllReduce(buf, buf, (size_t)n, 8, 0, c->comm, (hipStream_t)stream);
}
}  // namespace

static void comm_release(lrvb_ctx* c) {
    if (c->comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(c->comm); c->comm = nullptr; }
}

extern "C" int lrvb_comm_unique_id(lrvb_comm_id* id_out) {
    if (!id_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(rccl_load());
    RCCL_TRY(g_rccl.GetUniqueId(id_out));
    return LRVB_OK;
}
extern "C" int lrvb_comm_init(lrvb_ctx* c, int world_size, int rank, const lrvb_comm_id* id) {
    LRVB_TRY(ctx_bind(c));
    if (!id || world_size < 1 || rank < 0 || rank >= world_size) LRVB_FAIL(LRVB_ERR_INVALID, "bad communicator arguments");
    if (c->comm) LRVB_FAIL(LRVB_ERR_STATE, "this context already has a communicator");
    LRVB_TRY(rccl_load());
    RCCL_TRY(g_rccl.CommInitRank(&c->comm, world_size, *id, rank));
    c->comm_world = world_size; c->comm_rank = rank;
    c->reduce_fn = native_reduce; c->reduce_user = c;              // every observation sum now goes through RCCL
    c->hres_valid = false;
    return LRVB_OK;
}
extern "C" int lrvb_comm_destroy(lrvb_ctx* c) {
    if (!c) LRVB_FAIL(LRVB_ERR_INVALID, "null context");
    if (!c->comm) return LRVB_OK;
    (void)hipSetDevice(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->reduce_fn == native_reduce) { c->reduce_fn = nullptr; c->reduce_user = nullptr; }
    RCCL_TRY(g_rccl.CommDestroy(c->comm));
    c->comm = nullptr; c->comm_world = 1; c->comm_rank = 0;
    c->hvp_pt_valid = false; c->hres_valid = false;
    return LRVB_OK;
}
extern "C" int lrvb_allreduce_hessian(lrvb_ctx* c, double* stats_dev, int64_t n) {
    LRVB_TRY(ctx_bind(c));
    if (!stats_dev || n < 0) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    if (!c->comm) LRVB_FAIL(LRVB_ERR_STATE, "no communicator: call lrvb_comm_init first");
    RCCL_TRY(native_reduce(c, stats_dev, n, (void*)c->stream));
    return LRVB_OK;
}

// value and d f / d eta at the current eta; per-observation lp, cw stored.  stats: [value | g_glm].
// reduce: hand [value | g_glm] to the sum-over-ranks hook (callers that reduce a larger buffer themselves, or that
// only want the per-observation l', pass false).
static int eval_grad_eta(lrvb_ctx* c, double* stats_dev, bool include_quad, bool reduce = true) {
    if (c->loss != LRVB_LOSS_NONE) {
        LRVB_TRY(launch_glm_pass(c, PASS_GRAD, c->eta.p + c->glm_off, nullptr, stats_dev + 1, stats_dev, true));
        if (reduce) LRVB_TRY(obs_reduce(c, stats_dev, 1 + c->P));
        LRVB_TRY(launch_scatter_glm(c, stats_dev + 1, c->g_eta.p));
    } else {
        HIP_TRY(hipMemsetAsync(stats_dev, 0, sizeof(double), c->stream));
        LRVB_TRY(launch_scatter_glm(c, nullptr, c->g_eta.p));
    }
    if (include_quad) LRVB_TRY(launch_quad_grad_value(c, c->eta.p, c->g_eta.p, stats_dev));
    return LRVB_OK;
}

static int ensure_dense_J(lrvb_ctx* c, const double* theta_dev) {
    LRVB_TRY(buf_reserve(c, c->Jdense, (size_t)c->V * (size_t)c->D));
    return launch_dense_jac(c, theta_dev, c->Jdense.p);
}

// g_free = J^T g_eta
static int grad_to_free(lrvb_ctx* c, const double* theta_dev, double* g_free_dev) {
    if (c->all_box) { EW(mul_kernel, c->D, c->j1.p, c->g_eta.p, g_free_dev); return LRVB_OK; }
    LRVB_TRY(ensure_dense_J(c, theta_dev));
    return launch_gemv(c, true, c->V, c->D, 1.0, c->Jdense.p, c->D, c->g_eta.p, 0.0, g_free_dev);
}

// out_eta (V) = H_eta u   using the cached per-observation curvature cw
static int heta_apply(lrvb_ctx* c, const double* u_vec, double* out_vec) {
    if (c->loss != LRVB_LOSS_NONE) {
        LRVB_TRY(buf_reserve(c, c->vtmp3, (size_t)(c->V > c->P ? c->V : c->P)));
        LRVB_TRY(launch_glm_pass(c, PASS_HVP_C, nullptr, u_vec + c->glm_off, c->vtmp3.p, nullptr, false));
        LRVB_TRY(obs_reduce(c, c->vtmp3.p, c->P));
        // vtmp3 holds the P-vector; scatter into out_vec
        LRVB_TRY(launch_scatter_glm(c, c->vtmp3.p, out_vec));
    } else {
        LRVB_TRY(launch_scatter_glm(c, nullptr, out_vec));
    }
    return launch_quad_hvp(c, u_vec, out_vec);
}

// the same pass with whatever per-observation coefficient sits in c->cw, quadratic term optional
static int heta_apply_coef(lrvb_ctx* c, const double* u_vec, double* out_vec, bool with_quad) {
    if (c->loss != LRVB_LOSS_NONE) {
        LRVB_TRY(buf_reserve(c, c->vtmp3, (size_t)(c->V > c->P ? c->V : c->P)));
        LRVB_TRY(launch_glm_pass(c, PASS_HVP_C, nullptr, u_vec + c->glm_off, c->vtmp3.p, nullptr, false));
        LRVB_TRY(obs_reduce(c, c->vtmp3.p, c->P));
        LRVB_TRY(launch_scatter_glm(c, c->vtmp3.p, out_vec));
    } else {
        LRVB_TRY(launch_scatter_glm(c, nullptr, out_vec));
    }
    return with_quad ? launch_quad_hvp(c, u_vec, out_vec) : LRVB_OK;
}

// full HVP at the current point.  Requires set_point + eval_grad_eta done (g_eta, cw valid).
static int hvp_apply(lrvb_ctx* c, const double* theta_dev, bool is_free, const double* v_dev, double* out_dev) {
    if (!is_free) return heta_apply(c, v_dev, out_dev);
    if (c->all_box) {
        EW(mul_kernel, c->D, c->j1.p, v_dev, c->vtmp.p);                   // u = J v
        LRVB_TRY(heta_apply(c, c->vtmp.p, c->vtmp2.p));                   // H_eta u
        EW(fma3_kernel, c->D, c->g_eta.p, c->j2.p, v_dev, c->j1.p, c->vtmp2.p, out_dev);
        return LRVB_OK;
    }
    // general layouts: dense J and dense third-order matrix (built by the caller once per point)
    LRVB_TRY(launch_gemv(c, false, c->V, c->D, 1.0, c->Jdense.p, c->D, v_dev, 0.0, c->vtmp.p));
    LRVB_TRY(heta_apply(c, c->vtmp.p, c->vtmp2.p));
    LRVB_TRY(launch_gemv(c, true, c->V, c->D, 1.0, c->Jdense.p, c->D, c->vtmp2.p, 0.0, out_dev));
    LRVB_TRY(launch_gemv(c, false, c->D, c->D, 1.0, c->Tdense.p, c->D, v_dev, 1.0, out_dev));
    return LRVB_OK;
}

static int prepare_general_hvp(lrvb_ctx* c, const double* theta_dev) {
    if (c->all_box) return LRVB_OK;
    LRVB_TRY(ensure_dense_J(c, theta_dev));
    LRVB_TRY(buf_reserve(c, c->Tdense, (size_t)c->D * (size_t)c->D));
    HIP_TRY(hipMemsetAsync(c->Tdense.p, 0, (size_t)c->D * (size_t)c->D * sizeof(double), c->stream));
    return launch_third_order(c, theta_dev, c->g_eta.p, c->Tdense.p);
}

// ---- the resident Hessian ------------------------------------------------------------------------------------------
static int hres_capture(lrvb_ctx* c, const double* H_dev, i64 ld, const double* theta_dev, const double* theta_host) {
    const i64 D = c->D;
    c->hres_valid = false;
    LRVB_TRY(buf_reserve(c, c->Hres, (size_t)D * (size_t)D));
    HIP_TRY(hipMemcpy2DAsync(c->Hres.p, (size_t)D * 8, H_dev, (size_t)ld * 8, (size_t)D * 8, (size_t)D, hipMemcpyDeviceToDevice, c->stream));
    if (theta_host) { c->hres_pt.assign(theta_host, theta_host + D); c->hres_pt_host = true; }
    else {
        LRVB_TRY(buf_reserve(c, c->hres_theta, (size_t)D));
        HIP_TRY(hipMemcpyAsync(c->hres_theta.p, theta_dev, (size_t)D * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        c->hres_pt_host = false;
    }
    c->hres_valid = true;
    return LRVB_OK;
}
// Is `free_in` (host) the point of the resident Hessian?  A build through a `_dev` entry point left its point on the device
// only: the first question costs one small comparison kernel and a flag read-back, after which the host copy answers.
static int hres_matches(lrvb_ctx* c, const double* free_in, i64 D, bool* match) {
    *match = false;
    if (!c->hres_valid || c->no_resident || !free_in || D != c->D) return LRVB_OK;
    if (!c->hres_pt_host) {
        LRVB_TRY(h2d(c, c->theta.p, free_in, (size_t)D));
        int* flag = reinterpret_cast<int*>(c->scal.p);
        HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), c->stream));
        EW(vec_differs_kernel, D, (const double*)c->theta.p, (const double*)c->hres_theta.p, flag);
        int differs = 0;
        HIP_TRY(hipMemcpyAsync(&differs, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (differs) return LRVB_OK;                  // (the resident matrix stays: its own point may come back)
        c->hres_pt.assign(free_in, free_in + D); c->hres_pt_host = true;
        *match = true;
        return LRVB_OK;
    }
    *match = (i64)c->hres_pt.size() == D && memcmp(c->hres_pt.data(), free_in, (size_t)D * sizeof(double)) == 0;
    return LRVB_OK;
}

// ---- Hessian build ---------------------------------------------------------------------
extern "C" int lrvb_stats_size(lrvb_ctx* c, int64_t* n) {
    if (!c || !n) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    if (c->loss == LRVB_LOSS_NONE) { *n = 2; return LRVB_OK; }
    *n = 1 + c->P + (i64)wsyrk_num_tiles(c->P) * WS_TILE * WS_TILE;
    return LRVB_OK;
}

// ---- Gaussian loss: the build needs no pass over X besides the SYRK ------------------------------------------------

static bool gauss_shortcut(const lrvb_ctx* c) {
    return c->loss == LRVB_LOSS_GAUSSIAN && wsyrk_fast_path(c) && c->P <= 4096;
}

static int hessian_partial(lrvb_ctx* c, const double* point_dev, bool is_free, double* stats_dev) {
    LRVB_TRY(data_ready(c));
    LRVB_TRY(set_point(c, point_dev, is_free));
    if (gauss_shortcut(c)) {
        LRVB_TRY(reserve_obs_vec(c, c->cw));
        LRVB_TRY(reserve_obs_vec(c, c->cyv));
        LRVB_TRY(buf_reserve(c, c->rvec, (size_t)c->P));
        const int n_cyy = (int)((c->N + 2047) / 2048);
        const int nb = (int)((c->P + WS_TILE - 1) / WS_TILE);
        const size_t npart = (size_t)2 * nb * nb * WS_TILE;
        LRVB_TRY(buf_reserve(c, c->work1, npart + (size_t)n_cyy));
        double* cyy_part = c->work1.p + npart;
        hipLaunchKernelGGL(gauss_coef_kernel, dim3((unsigned)n_cyy), dim3(256), 0, c->stream, c->N, c->lik_info,
                           (const double*)c->w.p, (const double*)c->y.p, c->cw.p, c->cyv.p, cyy_part);
        HIP_TRY(hipGetLastError());
        LRVB_TRY(launch_wsyrk_r(c, c->cw.p, stats_dev + 1 + c->P, c->cyv.p, c->rvec.p));
        HIP_TRY(hipMemsetAsync(c->work1.p, 0, npart * sizeof(double), c->stream));
        const double* beta = c->eta.p + c->glm_off;
        hipLaunchKernelGGL(tiles_symv_kernel, dim3((unsigned)(nb * (nb + 1) / 2)), dim3(1024), 0, c->stream,
                           (const double*)(stats_dev + 1 + c->P), nb, c->P, beta, c->work1.p);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(gauss_finish_kernel, dim3(1), dim3(1024), 0, c->stream, (const double*)c->work1.p, nb, c->P,
                           beta, (const double*)c->rvec.p, (const double*)cyy_part, n_cyy, stats_dev, stats_dev + 1);
        HIP_TRY(hipGetLastError());
        return LRVB_OK;
    }
    LRVB_TRY(eval_grad_eta(c, stats_dev, false, false));
    if (c->loss != LRVB_LOSS_NONE)
        LRVB_TRY(launch_wsyrk(c, c->cw.p, stats_dev + 1 + c->P));
    return LRVB_OK;
}

static int hessian_finish(lrvb_ctx* c, const double* point_dev, bool is_free, const double* stats_dev,
                          double* H_dev, i64 ld) {
    const i64 n = is_free ? c->D : c->V;
    if (ld < n) LRVB_FAIL(LRVB_ERR_SIZE, "leading dimension %lld < %lld", (long long)ld, (long long)n);
    LRVB_TRY(set_point(c, point_dev, is_free));
    const double* tiles = (c->loss != LRVB_LOSS_NONE) ? stats_dev + 1 + c->P : nullptr;
    LRVB_TRY(launch_scatter_glm(c, (c->loss != LRVB_LOSS_NONE) ? stats_dev + 1 : nullptr, c->g_eta.p));
    LRVB_TRY(launch_quad_grad_value(c, c->eta.p, c->g_eta.p, nullptr));
    if (!is_free) {
        if (ld == c->V) return launch_build_Heta(c, tiles, H_dev);
        LRVB_TRY(buf_reserve(c, c->Heta, (size_t)c->V * (size_t)c->V));
        LRVB_TRY(launch_build_Heta(c, tiles, c->Heta.p));
        HIP_TRY(hipMemcpy2DAsync(H_dev, (size_t)ld * 8, c->Heta.p, (size_t)c->V * 8, (size_t)c->V * 8, (size_t)c->V, hipMemcpyDeviceToDevice, c->stream));
        return LRVB_OK;
    }
    if (c->all_box)
        return launch_finish_box(c, tiles, c->g_eta.p, c->j1.p, c->j2.p, true, H_dev, ld);
    // general: H = J^T H_eta J + T
    LRVB_TRY(buf_reserve(c, c->Heta, (size_t)c->V * (size_t)c->V));
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)c->V * (size_t)c->D));
    LRVB_TRY(launch_build_Heta(c, tiles, c->Heta.p));
    if (c->jt_rows > 0) {
        // box and log-Cholesky blocks: two structured products instead of the dense Jacobian and two V^2 D products
        LRVB_TRY(launch_jt_apply(c, point_dev, c->Heta.p, c->V, c->V, c->work1.p, c->V, false));      // W = J^T H_eta (D x V)
        if (ld == c->D) {
            LRVB_TRY(launch_jt_apply(c, point_dev, c->work1.p, c->V, c->D, H_dev, ld, true));         // J^T W^T
            return launch_third_order(c, point_dev, c->g_eta.p, H_dev);
        }
        LRVB_TRY(buf_reserve(c, c->Tdense, (size_t)c->D * (size_t)c->D));
        LRVB_TRY(launch_jt_apply(c, point_dev, c->work1.p, c->V, c->D, c->Tdense.p, c->D, true));
        LRVB_TRY(launch_third_order(c, point_dev, c->g_eta.p, c->Tdense.p));
        HIP_TRY(hipMemcpy2DAsync(H_dev, (size_t)ld * 8, c->Tdense.p, (size_t)c->D * 8, (size_t)c->D * 8, (size_t)c->D, hipMemcpyDeviceToDevice, c->stream));
        return LRVB_OK;
    }
    LRVB_TRY(ensure_dense_J(c, point_dev));
    LRVB_TRY(launch_gemm(c, false, false, c->V, c->D, c->V, 1.0, c->Heta.p, c->V, c->Jdense.p, c->D, 0.0, c->work1.p, c->D));
    // T into H_dev (respecting ld), then H += J^T work1
    if (ld == c->D) {
        HIP_TRY(hipMemsetAsync(H_dev, 0, (size_t)c->D * (size_t)c->D * sizeof(double), c->stream));
        LRVB_TRY(launch_third_order(c, point_dev, c->g_eta.p, H_dev));
        return launch_gemm(c, true, false, c->D, c->D, c->V, 1.0, c->Jdense.p, c->D, c->work1.p, c->D, 1.0, H_dev, ld);
    }
    LRVB_TRY(buf_reserve(c, c->Tdense, (size_t)c->D * (size_t)c->D));
    HIP_TRY(hipMemsetAsync(c->Tdense.p, 0, (size_t)c->D * (size_t)c->D * sizeof(double), c->stream));
    LRVB_TRY(launch_third_order(c, point_dev, c->g_eta.p, c->Tdense.p));
    LRVB_TRY(launch_gemm(c, true, false, c->D, c->D, c->V, 1.0, c->Jdense.p, c->D, c->work1.p, c->D, 1.0, c->Tdense.p, c->D));
    HIP_TRY(hipMemcpy2DAsync(H_dev, (size_t)ld * 8, c->Tdense.p, (size_t)c->D * 8, (size_t)c->D * 8, (size_t)c->D, hipMemcpyDeviceToDevice, c->stream));
    return LRVB_OK;
}

// the whole statistics buffer [value | g_glm | tiles] of a build in ONE reduction
static int stats_reduce(lrvb_ctx* c) {
    if (!c->reduce_fn || c->loss == LRVB_LOSS_NONE) return LRVB_OK;
    int64_t n = 0;
    LRVB_TRY(lrvb_stats_size(c, &n));
    return obs_reduce(c, c->stats.p, n);
}

extern "C" int lrvb_hessian_partial_dev(lrvb_ctx* c, const double* free_dev, double* stats_dev) {
    LRVB_TRY(ctx_bind(c));
    if (!free_dev || !stats_dev) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    return hessian_partial(c, free_dev, true, stats_dev);
}
extern "C" int lrvb_hessian_finish_dev(lrvb_ctx* c, const double* free_dev, const double* stats_dev, double* H_dev, int64_t ld) {
    LRVB_TRY(ctx_bind(c));
    if (!free_dev || !stats_dev || !H_dev) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(hessian_finish(c, free_dev, true, stats_dev, H_dev, ld));
    return hres_capture(c, H_dev, ld, free_dev, nullptr);
}
extern "C" int lrvb_hessian_dev(lrvb_ctx* c, const double* free_dev, double* H_dev, int64_t ld) {
    LRVB_TRY(ctx_bind(c));
    if (!free_dev || !H_dev) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_BUILD));
    LRVB_TRY(hessian_partial(c, free_dev, true, c->stats.p));
    LRVB_TRY(stats_reduce(c));
    LRVB_TRY(hessian_finish(c, free_dev, true, c->stats.p, H_dev, ld));
    LRVB_TRY(hres_capture(c, H_dev, ld, free_dev, nullptr));
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_BUILD));
    return LRVB_OK;
}

static int check_len(i64 got, i64 want, const char* what) {
    if (got != want) LRVB_FAIL(LRVB_ERR_SIZE, "Wrong size for %s.  Expected %lld, got %lld", what, (long long)want, (long long)got);
    return LRVB_OK;
}

static int hessian_host(lrvb_ctx* c, const double* point, i64 n_in, bool is_free, double* H_out, i64 ld) {
    LRVB_TRY(ctx_bind(c));
    if (!point || !H_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    const i64 n = is_free ? c->D : c->V;
    LRVB_TRY(check_len(n_in, n, is_free ? "free vector" : "vector"));
    if (ld < n) LRVB_FAIL(LRVB_ERR_SIZE, "leading dimension too small");
    LRVB_TRY(h2d(c, c->theta.p, point, (size_t)n));           // theta buffer doubles as the vector-mode input
    LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)n * (size_t)n));
    LRVB_TRY(hessian_partial(c, c->theta.p, is_free, c->stats.p));
    LRVB_TRY(stats_reduce(c));
    LRVB_TRY(hessian_finish(c, c->theta.p, is_free, c->stats.p, c->Hfree.p, n));
    if (is_free) LRVB_TRY(hres_capture(c, c->Hfree.p, n, c->theta.p, point));
    HIP_TRY(hipMemcpy2DAsync(H_out, (size_t)ld * 8, c->Hfree.p, (size_t)n * 8, (size_t)n * 8, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LRVB_OK;
}
extern "C" int lrvb_hessian(lrvb_ctx* c, const double* free_in, int64_t D, double* H_out, int64_t ld) {
    return hessian_host(c, free_in, D, true, H_out, ld);
}
extern "C" int lrvb_hessian_vec(lrvb_ctx* c, const double* vec_in, int64_t V, double* H_out, int64_t ld) {
    return hessian_host(c, vec_in, V, false, H_out, ld);
}

// ---- value / gradient ------------------------------------------------------------------
// Is (point, is_free) the point whose state the last value / gradient / HVP call left in place?  Must be asked
// BEFORE ctx_bind, which clears the flag for every entry point.
static bool same_point(const lrvb_ctx* c, const double* point, i64 n_in, bool is_free) {
    return c && point && c->hvp_pt_valid && c->hvp_pt_free == is_free && (i64)c->hvp_pt.size() == n_in &&
           memcmp(c->hvp_pt.data(), point, (size_t)n_in * sizeof(double)) == 0;
}
static void remember_point(lrvb_ctx* c, const double* point, i64 n, bool is_free, bool prepared) {
    c->hvp_pt.assign(point, point + n);
    c->hvp_pt_free = is_free;
    c->hvp_pt_prepared = prepared;
    c->hvp_pt_valid = true;
}

static int grad_host(lrvb_ctx* c, const double* point, i64 n_in, bool is_free, double* value_out, double* g_out) {
    const bool reuse = same_point(c, point, n_in, is_free);          // fun(x) then jac(x), as scipy calls them
    const bool prepared = reuse && c->hvp_pt_prepared;
    LRVB_TRY(ctx_bind(c));
    if (!point) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    const i64 n = is_free ? c->D : c->V;
    LRVB_TRY(check_len(n_in, n, is_free ? "free vector" : "vector"));
    LRVB_TRY(data_ready(c));
    if (!reuse) {
        LRVB_TRY(h2d(c, c->theta.p, point, (size_t)n));
        LRVB_TRY(set_point(c, c->theta.p, is_free));
        LRVB_TRY(eval_grad_eta(c, c->stats.p, true));
    }
    if (g_out) {
        if (is_free) { LRVB_TRY(grad_to_free(c, c->theta.p, c->g_free.p)); LRVB_TRY(d2h(c, g_out, c->g_free.p, (size_t)n)); }
        else LRVB_TRY(d2h(c, g_out, c->g_eta.p, (size_t)n));
    }
    if (value_out) LRVB_TRY(d2h(c, value_out, c->stats.p, 1));
    remember_point(c, point, n, is_free, prepared);       // (grad_to_free rebuilds the same dense Jacobian)
    return LRVB_OK;
}
extern "C" int lrvb_value(lrvb_ctx* c, const double* free_in, int64_t D, double* out) {
    if (!out) LRVB_FAIL(LRVB_ERR_INVALID, "null out");
    return grad_host(c, free_in, D, true, out, nullptr);
}
extern "C" int lrvb_grad(lrvb_ctx* c, const double* free_in, int64_t D, double* value_out, double* g_out) {
    if (!g_out) LRVB_FAIL(LRVB_ERR_INVALID, "null out");
    return grad_host(c, free_in, D, true, value_out, g_out);
}
extern "C" int lrvb_value_vec(lrvb_ctx* c, const double* vec_in, int64_t V, double* out) {
    if (!out) LRVB_FAIL(LRVB_ERR_INVALID, "null out");
    return grad_host(c, vec_in, V, false, out, nullptr);
}
extern "C" int lrvb_grad_vec(lrvb_ctx* c, const double* vec_in, int64_t V, double* value_out, double* g_out) {
    if (!g_out) LRVB_FAIL(LRVB_ERR_INVALID, "null out");
    return grad_host(c, vec_in, V, false, value_out, g_out);
}

// ---- HVP --------------------------------------------------------------------------------
static int hvp_dev_impl(lrvb_ctx* c, const double* point_dev, bool is_free, const double* v_dev, double* out_dev) {
    LRVB_TRY(data_ready(c));
    LRVB_TRY(set_point(c, point_dev, is_free));
    LRVB_TRY(eval_grad_eta(c, c->stats.p, true));
    if (is_free) LRVB_TRY(prepare_general_hvp(c, point_dev));
    return hvp_apply(c, point_dev, is_free, v_dev, out_dev);
}
extern "C" int lrvb_hvp_dev(lrvb_ctx* c, const double* free_dev, const double* v_dev, double* out_dev) {
    LRVB_TRY(ctx_bind(c));
    if (!free_dev || !v_dev || !out_dev) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    return hvp_dev_impl(c, free_dev, true, v_dev, out_dev);
}
// Many products at ONE point (scipy's trust-ncg or cg driving lrvb_hvp, right-hand sides solved one by one): a pass over X per
// product is the cheaper route for a handful; a build of the point's Hessian costs about D / 86 passes and makes every further
// product a D x D matrix-vector product.  Past max(8, D / 64) matrix-free products at the remembered point the Hessian is
// built into the resident slot (as lrvb_hessian would leave it) and *resident set.  The point state must be current
// (c->theta on the device, set_point + eval_grad_eta done); every rank counts the same products, so a sharded run builds
// on all ranks at the same product.
static int maybe_build_resident(lrvb_ctx* c, const double* point_host, i64 D, bool* resident) {
    if (*resident || c->no_resident || c->loss == LRVB_LOSS_NONE || D < 256 || D > 8192) return LRVB_OK;
    const i64 thr = D / 64 > 8 ? D / 64 : 8;
    if (c->pt_products <= thr) return LRVB_OK;
    c->hres_valid = false;
    LRVB_TRY(buf_reserve(c, c->Hres, (size_t)D * (size_t)D));
    LRVB_TRY(hessian_partial(c, c->theta.p, true, c->stats.p));
    LRVB_TRY(stats_reduce(c));
    LRVB_TRY(hessian_finish(c, c->theta.p, true, c->stats.p, c->Hres.p, D));
    c->hres_pt.assign(point_host, point_host + D); c->hres_pt_host = true;
    c->hres_valid = true;
    *resident = true;
    return LRVB_OK;
}

static int hvp_host(lrvb_ctx* c, const double* point, const double* v, i64 n_in, bool is_free, double* out) {
    // same point as the previous call, nothing else in between: its eta / J / g_eta / curvature are still in place
    const bool reuse = same_point(c, point, n_in, is_free);
    const bool prepared = reuse && c->hvp_pt_prepared;
    LRVB_TRY(ctx_bind(c));
    if (!point || !v || !out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    const i64 n = is_free ? c->D : c->V;
    LRVB_TRY(check_len(n_in, n, is_free ? "free vector" : "vector"));
    LRVB_TRY(data_ready(c));
    LRVB_TRY(buf_reserve(c, c->cgp, (size_t)n));
    LRVB_TRY(buf_reserve(c, c->cgq, (size_t)n));
    if (is_free) {                                        // the Hessian of this very point is resident: one D x D product
        bool resident = false;
        LRVB_TRY(hres_matches(c, point, n_in, &resident));
        if (resident) {
            LRVB_TRY(h2d(c, c->cgp.p, v, (size_t)n));
            LRVB_TRY(launch_gemv(c, false, n, n, 1.0, c->Hres.p, n, c->cgp.p, 0.0, c->cgq.p));
            return d2h(c, out, c->cgq.p, (size_t)n);
        }
    }
    LRVB_TRY(h2d(c, c->cgp.p, v, (size_t)n));
    if (!reuse) {
        LRVB_TRY(h2d(c, c->theta.p, point, (size_t)n));
        LRVB_TRY(set_point(c, c->theta.p, is_free));
        LRVB_TRY(eval_grad_eta(c, c->stats.p, true));
        c->pt_products = 0;
    }
    if (is_free) {
        ++c->pt_products;
        bool built = false;
        LRVB_TRY(maybe_build_resident(c, point, n, &built));
        if (built) {
            LRVB_TRY(launch_gemv(c, false, n, n, 1.0, c->Hres.p, n, c->cgp.p, 0.0, c->cgq.p));
            return d2h(c, out, c->cgq.p, (size_t)n);
        }
    }
    if (!prepared && is_free) LRVB_TRY(prepare_general_hvp(c, c->theta.p));
    LRVB_TRY(hvp_apply(c, c->theta.p, is_free, c->cgp.p, c->cgq.p));
    LRVB_TRY(d2h(c, out, c->cgq.p, (size_t)n));
    remember_point(c, point, n, is_free, true);
    return LRVB_OK;
}
extern "C" int lrvb_hvp(lrvb_ctx* c, const double* free_in, const double* v, int64_t D, double* out) {
    return hvp_host(c, free_in, v, D, true, out);
}
extern "C" int lrvb_hvp_vec(lrvb_ctx* c, const double* vec_in, const double* v, int64_t V, double* out) {
    return hvp_host(c, vec_in, v, V, false, out);
}

// ---- packing entry points ----------------------------------------------------------------
extern "C" int lrvb_constrain(lrvb_ctx* c, const double* free_in, int64_t D, double* vec_out, int64_t V) {
    LRVB_TRY(ctx_bind(c));
    if (!free_in || !vec_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(check_len(D, c->D, "free vector")); LRVB_TRY(check_len(V, c->V, "vector"));
    LRVB_TRY(h2d(c, c->theta.p, free_in, (size_t)D));
    LRVB_TRY(launch_constrain(c, c->theta.p, c->eta.p, c->j1.p, c->j2.p));
    return d2h(c, vec_out, c->eta.p, (size_t)V);
}
extern "C" int lrvb_unconstrain(lrvb_ctx* c, const double* vec_in, int64_t V, double* free_out, int64_t D) {
    LRVB_TRY(ctx_bind(c));
    if (!vec_in || !free_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(check_len(D, c->D, "free vector")); LRVB_TRY(check_len(V, c->V, "vector"));
    LRVB_TRY(h2d(c, c->eta.p, vec_in, (size_t)V));
    int* flag = reinterpret_cast<int*>(c->scal.p);
    HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), c->stream));
    LRVB_TRY(launch_unconstrain(c, c->eta.p, c->theta.p, flag));
    int bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (bad & 1) LRVB_FAIL(LRVB_ERR_INVALID, "Elements outside the bounds");
    if (bad & 2) LRVB_FAIL(LRVB_ERR_INVALID, "Matrix is not positive definite above diag_lb");
    return d2h(c, free_out, c->theta.p, (size_t)D);
}
extern "C" int lrvb_free_to_vector_jac(lrvb_ctx* c, const double* free_in, int64_t D, double* jac_out) {
    LRVB_TRY(ctx_bind(c));
    if (!free_in || !jac_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(check_len(D, c->D, "free vector"));
    LRVB_TRY(h2d(c, c->theta.p, free_in, (size_t)D));
    LRVB_TRY(ensure_dense_J(c, c->theta.p));
    return d2h(c, jac_out, c->Jdense.p, (size_t)c->V * (size_t)c->D);
}
extern "C" int lrvb_free_hessian_from_vector(lrvb_ctx* c, const double* free_in, const double* g_vec,
                                             const double* H_vec, double* H_free_out) {
    LRVB_TRY(ctx_bind(c));
    if (!free_in || !g_vec || !H_vec || !H_free_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    const i64 D = c->D, V = c->V;
    LRVB_TRY(h2d(c, c->theta.p, free_in, (size_t)D));
    LRVB_TRY(h2d(c, c->g_eta.p, g_vec, (size_t)V));
    LRVB_TRY(buf_reserve(c, c->Heta, (size_t)V * (size_t)V));
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)V * (size_t)D));
    LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)D * (size_t)D));
    LRVB_TRY(h2d(c, c->Heta.p, H_vec, (size_t)V * (size_t)V));
    if (c->jt_rows > 0) {                                  // box and log-Cholesky blocks: the structured products (k_pack.hip).
        // H_vec is taken as given (J^T H_vec J also for a matrix that is not exactly symmetric): both products read their
        // operand transposed -- W = J^T H_vec^T = (H_vec J)^T, then J^T W^T = J^T H_vec J
        LRVB_TRY(launch_jt_apply(c, c->theta.p, c->Heta.p, V, V, c->work1.p, V, true));             // (H_vec J)^T    (D x V)
        LRVB_TRY(launch_jt_apply(c, c->theta.p, c->work1.p, V, D, c->Hfree.p, D, true));            // J^T (H_vec J)
        LRVB_TRY(launch_third_order(c, c->theta.p, c->g_eta.p, c->Hfree.p));
        return d2h(c, H_free_out, c->Hfree.p, (size_t)D * (size_t)D);
    }
    LRVB_TRY(ensure_dense_J(c, c->theta.p));
    LRVB_TRY(launch_gemm(c, false, false, V, D, V, 1.0, c->Heta.p, V, c->Jdense.p, D, 0.0, c->work1.p, D));
    HIP_TRY(hipMemsetAsync(c->Hfree.p, 0, (size_t)D * (size_t)D * sizeof(double), c->stream));
    LRVB_TRY(launch_third_order(c, c->theta.p, c->g_eta.p, c->Hfree.p));
    LRVB_TRY(launch_gemm(c, true, false, D, D, V, 1.0, c->Jdense.p, D, c->work1.p, D, 1.0, c->Hfree.p, D));
    return d2h(c, H_free_out, c->Hfree.p, (size_t)D * (size_t)D);
}

static int gemm_tn(lrvb_ctx* c, i64 K, i64 PA, i64 PB, const double* A, const double* B, double* C);

// ---- vector-coordinate Hessian assembled on the device from small host blocks ------------------------------
extern "C" int lrvb_hvec_begin(lrvb_ctx* c) {
    LRVB_TRY(ctx_bind(c));
    LRVB_TRY(buf_reserve(c, c->Heta, (size_t)c->V * (size_t)c->V));
    HIP_TRY(hipMemsetAsync(c->Heta.p, 0, (size_t)c->V * (size_t)c->V * sizeof(double), c->stream));
    c->hvec_open = true;
    return LRVB_OK;
}
extern "C" int lrvb_hvec_add_block(lrvb_ctx* c, const double* block, int64_t rows, int64_t cols, int64_t row_off,
                                   int64_t col_off, int mirror) {
    LRVB_TRY(ctx_bind(c));
    if (!c->hvec_open) LRVB_FAIL(LRVB_ERR_STATE, "call lrvb_hvec_begin first");
    if (!block || rows <= 0 || cols <= 0 || row_off < 0 || col_off < 0 || row_off + rows > c->V || col_off + cols > c->V)
        LRVB_FAIL(LRVB_ERR_INVALID, "block [%lld+%lld, %lld+%lld) outside the %lld x %lld matrix", (long long)row_off, (long long)rows,
                  (long long)col_off, (long long)cols, (long long)c->V, (long long)c->V);
    if (mirror && row_off == col_off) LRVB_FAIL(LRVB_ERR_INVALID, "a mirrored block cannot sit on the diagonal");
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)(rows * cols)));
    LRVB_TRY(h2d(c, c->work1.p, block, (size_t)(rows * cols)));
    EW(hvec_add_block_kernel, rows * cols, cols, c->work1.p, c->Heta.p, c->V, row_off, col_off, mirror);
    return LRVB_OK;
}
extern "C" int lrvb_hvec_add_indexed(lrvb_ctx* c, const double* block, int64_t nr, int64_t nc, const int64_t* rows, const int64_t* cols) {
    LRVB_TRY(ctx_bind(c));
    if (!c->hvec_open) LRVB_FAIL(LRVB_ERR_STATE, "call lrvb_hvec_begin first");
    if (!block || !rows || !cols || nr <= 0 || nc <= 0) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    std::vector<double> pack((size_t)(nr * nc + nr + nc));
    memcpy(pack.data(), block, (size_t)(nr * nc) * sizeof(double));
    for (i64 a = 0; a < nr; ++a) {
        if (rows[a] < 0 || rows[a] >= c->V) LRVB_FAIL(LRVB_ERR_INVALID, "row index %lld outside [0, %lld)", (long long)rows[a], (long long)c->V);
        pack[(size_t)(nr * nc + a)] = (double)rows[a];
    }
    for (i64 b = 0; b < nc; ++b) {
        if (cols[b] < 0 || cols[b] >= c->V) LRVB_FAIL(LRVB_ERR_INVALID, "column index %lld outside [0, %lld)", (long long)cols[b], (long long)c->V);
        pack[(size_t)(nr * nc + nr + b)] = (double)cols[b];
    }
    // two entries of one index list must not name the same element twice (the += would race): lists of a parameter layout never do
    LRVB_TRY(buf_reserve(c, c->work1, pack.size()));
    LRVB_TRY(h2d(c, c->work1.p, pack.data(), pack.size()));
    EW(hvec_add_indexed_kernel, nr * nc, nc, (const double*)c->work1.p, (const double*)(c->work1.p + nr * nc),
       (const double*)(c->work1.p + nr * nc + nr), c->Heta.p, c->V);
    return LRVB_OK;
}
extern "C" int lrvb_hvec_add_symkron(lrvb_ctx* c, const double* A, const double* B, int64_t k, double coef,
                                     int64_t row_off, int64_t col_off, int mirror) {
    LRVB_TRY(ctx_bind(c));
    if (!c->hvec_open) LRVB_FAIL(LRVB_ERR_STATE, "call lrvb_hvec_begin first");
    const i64 m = k * (k + 1) / 2;
    if (!A || !B || k <= 0 || k > 2048 || row_off < 0 || col_off < 0 || row_off + m > c->V || col_off + m > c->V)
        LRVB_FAIL(LRVB_ERR_INVALID, "Kronecker block of order %lld at (%lld, %lld) outside the %lld x %lld matrix", (long long)m,
                  (long long)row_off, (long long)col_off, (long long)c->V, (long long)c->V);
    if (mirror && row_off == col_off) LRVB_FAIL(LRVB_ERR_INVALID, "a mirrored block cannot sit on the diagonal");
    LRVB_TRY(buf_reserve(c, c->vtmp3, (size_t)(2 * k * k)));
    LRVB_TRY(h2d(c, c->vtmp3.p, A, (size_t)(k * k)));
    LRVB_TRY(h2d(c, c->vtmp3.p + k * k, B, (size_t)(k * k)));
    EW(hvec_symkron_kernel, m * m, m, (int)k, c->vtmp3.p, c->vtmp3.p + k * k, coef, c->Heta.p, c->V, row_off, col_off, mirror);
    return LRVB_OK;
}
// H_free = J^T H_vec J + sum_k g_k d2 eta_k with H_vec the matrix assembled by the lrvb_hvec_* calls.  The result
// stays on the device (lrvb_chol_factor_last factors it); H_free_out may be NULL.  is_free = 0 returns H_vec itself.
static int hvec_finish_impl(lrvb_ctx* c, const double* point, int64_t n_in, int is_free, const double* g_vec, double* H_out);
extern "C" int lrvb_hvec_finish(lrvb_ctx* c, const double* point, int64_t n_in, int is_free, const double* g_vec, double* H_out) {
    LRVB_TRY(ctx_bind(c));
    if (!c->hvec_open) LRVB_FAIL(LRVB_ERR_STATE, "call lrvb_hvec_begin first");
    return hvec_finish_impl(c, point, n_in, is_free, g_vec, H_out);
}
// begin + every block + finish in one call: the operands travel in ONE upload, the blocks are launches in stream order
extern "C" int lrvb_hvec_program(lrvb_ctx* c, const int64_t* ops, int64_t n_ops, const double* data, int64_t n_data,
                                 const double* point, int64_t n_in, int is_free, const double* g_vec, double* H_out) {
    LRVB_TRY(ctx_bind(c));
    if (n_ops < 0 || n_data < 0 || (n_ops > 0 && !ops) || (n_data > 0 && !data)) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    const i64 V = c->V;
    // every record is checked before anything is launched
    for (i64 t = 0; t < n_ops; ++t) {
        const int64_t* o = ops + 8 * t;
        const i64 kind = o[0], off = o[1], a = o[2], b = o[3], ro = o[4], co = o[5], mir = o[6], f = o[7];
        if (off < 0) LRVB_FAIL(LRVB_ERR_INVALID, "record %lld: negative operand offset", (long long)t);
        if (kind == 0) {
            if (a <= 0 || b <= 0 || off + a * b > n_data || ro < 0 || co < 0 || ro + a > V || co + b > V || (mir && ro == co))
                LRVB_FAIL(LRVB_ERR_INVALID, "record %lld: block [%lld+%lld, %lld+%lld) does not fit the %lld x %lld matrix or its operand the data", (long long)t,
                          (long long)ro, (long long)a, (long long)co, (long long)b, (long long)V, (long long)V);
        } else if (kind == 1) {
            if (a <= 0 || b <= 0 || off + a * b + a + b > n_data) LRVB_FAIL(LRVB_ERR_INVALID, "record %lld: indexed block outside the data", (long long)t);
            for (i64 e = 0; e < a + b; ++e) {
                const double ix = data[off + a * b + e];
                if (!(ix >= 0.0) || ix >= (double)V || ix != (double)(i64)ix) LRVB_FAIL(LRVB_ERR_INVALID, "record %lld: index %g outside [0, %lld)", (long long)t, ix, (long long)V);
            }
        } else if (kind == 2) {
            const i64 m = a * (a + 1) / 2;
            if (a <= 0 || a > 2048 || off + a * a > n_data || f < 0 || f + a * a > n_data || b < 0 || b >= n_data || ro < 0 || co < 0 ||
                ro + m > V || co + m > V || (mir && ro == co))
                LRVB_FAIL(LRVB_ERR_INVALID, "record %lld: Kronecker block of order %lld at (%lld, %lld) does not fit", (long long)t, (long long)m, (long long)ro, (long long)co);
        } else LRVB_FAIL(LRVB_ERR_INVALID, "record %lld: unknown kind %lld", (long long)t, (long long)kind);
    }
    LRVB_TRY(buf_reserve(c, c->Heta, (size_t)V * (size_t)V));
    HIP_TRY(hipMemsetAsync(c->Heta.p, 0, (size_t)V * (size_t)V * sizeof(double), c->stream));
    if (n_data > 0) {
        LRVB_TRY(buf_reserve(c, c->hprog, (size_t)n_data));
        LRVB_TRY(h2d(c, c->hprog.p, data, (size_t)n_data));
    }
    const double* dd = c->hprog.p;
    for (i64 t = 0; t < n_ops; ++t) {
        const int64_t* o = ops + 8 * t;
        const i64 kind = o[0], off = o[1], a = o[2], b = o[3], ro = o[4], co = o[5], f = o[7];
        const int mir = o[6] != 0;
        if (kind == 0) {
            EW(hvec_add_block_kernel, a * b, b, dd + off, c->Heta.p, V, ro, co, mir);
        } else if (kind == 1) {
            EW(hvec_add_indexed_kernel, a * b, b, dd + off, dd + off + a * b, dd + off + a * b + a, c->Heta.p, V);
        } else {
            const i64 m = a * (a + 1) / 2;
            EW(hvec_symkron_kernel, m * m, m, (int)a, dd + off, dd + f, data[b], c->Heta.p, V, ro, co, mir);
        }
    }
    c->hvec_open = true;
    return hvec_finish_impl(c, point, n_in, is_free, g_vec, H_out);
}
static int hvec_finish_impl(lrvb_ctx* c, const double* point, int64_t n_in, int is_free, const double* g_vec, double* H_out) {
    c->hvec_open = false;
    const i64 D = c->D, V = c->V;
    if (!is_free) {
        LRVB_TRY(check_len(n_in, V, "vector"));
        return H_out ? d2h(c, H_out, c->Heta.p, (size_t)V * (size_t)V) : LRVB_OK;
    }
    if (!point || !g_vec) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(check_len(n_in, D, "free vector"));
    LRVB_TRY(h2d(c, c->theta.p, point, (size_t)D));
    LRVB_TRY(h2d(c, c->g_eta.p, g_vec, (size_t)V));
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)V * (size_t)D));
    LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)D * (size_t)D));
    LRVB_TRY(buf_reserve(c, c->Tdense, (size_t)D * (size_t)D));
    if (c->jt_rows > 0) {                                  // box and log-Cholesky blocks: the structured products (k_pack.hip)
        LRVB_TRY(launch_jt_apply(c, c->theta.p, c->Heta.p, V, V, c->work1.p, V, false));            // J^T H_vec (H_vec is symmetric by construction)
        LRVB_TRY(launch_jt_apply(c, c->theta.p, c->work1.p, V, D, c->Hfree.p, D, true));
        LRVB_TRY(launch_third_order(c, c->theta.p, c->g_eta.p, c->Hfree.p));
        return H_out ? d2h(c, H_out, c->Hfree.p, (size_t)D * (size_t)D) : LRVB_OK;
    }
    LRVB_TRY(ensure_dense_J(c, c->theta.p));
    LRVB_TRY(gemm_tn(c, V, V, D, c->Heta.p, c->Jdense.p, c->work1.p));          // H_vec is symmetric: H J = H^T J
    LRVB_TRY(gemm_tn(c, V, D, D, c->Jdense.p, c->work1.p, c->Tdense.p));        // J^T (H J)
    HIP_TRY(hipMemsetAsync(c->Hfree.p, 0, (size_t)D * (size_t)D * sizeof(double), c->stream));
    LRVB_TRY(launch_third_order(c, c->theta.p, c->g_eta.p, c->Hfree.p));
    LRVB_TRY(launch_axpby(c, D * D, 1.0, c->Tdense.p, 1.0, c->Hfree.p));
    return H_out ? d2h(c, H_out, c->Hfree.p, (size_t)D * (size_t)D) : LRVB_OK;
}

// ---- configurations 2 and 4 as ONE call: statistics, closed forms, assembly, free conversion -- no host round trip ------------
static int grouped_stats_device(lrvb_ctx* c);
static int lmm_group_terms_device(lrvb_ctx* c, const double* par, int64_t n_par, const double* f_local, int64_t n_local, double** sums_dev);
struct LmmTermsLayout { double *Cm, *wts, *dpar, *dloc, *part, *sums, *Md; int ldc; i64 grid, n_waves, G, p, R; };
static int lmm_group_terms_prepare(lrvb_ctx* c, const double* par, int64_t n_par, const double* f_local, int64_t n_local, LmmTermsLayout& L);
static int lmm_group_terms_launch(lrvb_ctx* c, const LmmTermsLayout& L, bool with_sums);
// J^T H_vec J + sum_k g_k d2 eta_k for the vector-coordinate matrix in c->Heta (leading dimension Vp = V rounded up to even,
// the padding zero) with theta on the device and g in c->g_eta; the result in c->Hfree (leading dimension D), where
// lrvb_chol_factor_last finds it.  Even widths throughout: the two products run on the LDS-DMA MFMA kernel without the padded
// copies of gemm_tn (three rectangular copies and a memset per product at the 995 parameters of configuration 4).
static int free_conversion_reserve(lrvb_ctx* c, i64 Vp) {
    const i64 D = c->D, Dp = D + (D & 1);
    LRVB_TRY(buf_reserve(c, c->Jdense, (size_t)Vp * (size_t)Dp));
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)Vp * (size_t)Dp));
    LRVB_TRY(buf_reserve(c, c->Tdense, (size_t)Dp * (size_t)Dp));
    return buf_reserve(c, c->Hfree, (size_t)D * (size_t)D);
}
// Clears c->Heta (Vp x Vp) and, where the conversion below builds the dense Jacobian, c->Jdense with it -- one launch.
static int free_conversion_clear(lrvb_ctx* c, i64 Vp) {
    const i64 Dp = c->D + (c->D & 1);
    return launch_zero2(c, c->Heta.p, (size_t)Vp * (size_t)Vp, c->Jdense.p, c->jt_rows > 0 ? 0 : (size_t)Vp * (size_t)Dp);
}
// (after free_conversion_reserve + free_conversion_clear and the assembly of c->Heta)
static int free_conversion_padded(lrvb_ctx* c, const double* theta_dev, i64 Vp) {
    const i64 D = c->D, Dp = D + (D & 1);
    if (c->jt_rows > 0) {
        // box and log-Cholesky blocks only: J^T H J as two structured products (k_pack.hip), no dense Jacobian
        LRVB_TRY(launch_jt_apply(c, theta_dev, c->Heta.p, Vp, Vp, c->work1.p, Vp, false));      // W = J^T H       (D x Vp)
        LRVB_TRY(launch_jt_apply(c, theta_dev, c->work1.p, Vp, D, c->Hfree.p, D, true));        // J^T W^T = J^T H J
        return launch_third_order(c, theta_dev, c->g_eta.p, c->Hfree.p);                        // ... + sum_k g_k d2 eta_k
    }
    LRVB_TRY(launch_dense_jac(c, theta_dev, c->Jdense.p, Dp, Vp, true));
    LRVB_TRY(gemm_tn(c, Vp, Vp, Dp, c->Heta.p, c->Jdense.p, c->work1.p));          // H_vec is symmetric: H J = H^T J
    if (Dp == D) {                                                                  // even D: J^T (H J) lands where the result lives
        LRVB_TRY(gemm_tn(c, Vp, Dp, Dp, c->Jdense.p, c->work1.p, c->Hfree.p));
    } else {
        LRVB_TRY(gemm_tn(c, Vp, Dp, Dp, c->Jdense.p, c->work1.p, c->Tdense.p));    // J^T (H J)
        LRVB_TRY(launch_add_padded(c, D, c->Tdense.p, Dp, c->Hfree.p, D));         // ... compacted to leading dimension D
    }
    return launch_third_order(c, theta_dev, c->g_eta.p, c->Hfree.p);                // ... + sum_k g_k d2 eta_k
}

extern "C" int lrvb_mvnreg_hessian(lrvb_ctx* c, const double* free_in, int64_t D, const double* hp, int64_t n_hp, const int32_t* idx,
                                   double* value_out, double* H_out) {
    LRVB_TRY(ctx_bind(c));
    if (!free_in || !hp || !idx) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(check_len(D, c->D, "free vector"));
    if (c->loss == LRVB_LOSS_NONE || !c->have_X) LRVB_FAIL(LRVB_ERR_STATE, "no data matrix: call lrvb_set_data(LRVB_SLOT_X) first");
    const i64 q = c->P, k = q - 1, V = c->V, Vp = V + (V & 1), mm = k * (k + 1) / 2;
    if (k < 1 || k > 63) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "1 <= k <= 63 regressors");
    LRVB_TRY(check_len(n_hp, 32 + 2 * k + 3 * k * k, "host pack"));
    MvnRegIdx ix{ (int)k, idx[0], idx[1], idx[2], idx[3], Vp };
    if (ix.ms < 0 || ix.ms + k > V || ix.ls < 0 || ix.ls + mm > V || ix.ia < 0 || ix.ia >= V || ix.ib < 0 || ix.ib >= V || V != k + mm + 2)
        LRVB_FAIL(LRVB_ERR_INVALID, "parameter positions do not describe [mean, information, shape, rate] of %lld + %lld + 2 vector coordinates", (long long)k, (long long)mm);
    // one upload: [theta | hp]
    std::vector<double> pack((size_t)(D + n_hp));
    memcpy(pack.data(), free_in, (size_t)D * sizeof(double));
    memcpy(pack.data() + D, hp, (size_t)n_hp * sizeof(double));
    LRVB_TRY(buf_reserve(c, c->hprog, pack.size()));
    LRVB_TRY(buf_reserve(c, c->qstats, (size_t)(q * q) + 1 + 256));
    LRVB_TRY(buf_reserve(c, c->Heta, (size_t)Vp * (size_t)Vp));
    LRVB_TRY(buf_reserve(c, c->vtmp3, (size_t)(3 * k * k + 1) > (size_t)(V > D ? V : D) ? (size_t)(3 * k * k + 1) : (size_t)(V > D ? V : D)));
    LRVB_TRY(free_conversion_reserve(c, Vp));
    LRVB_TRY(h2d(c, c->hprog.p, pack.data(), pack.size()));
    const double* hp_dev = c->hprog.p + D;
    double* scratch = c->vtmp3.p; double* Gc = scratch + 2 * k * k; double* val = Gc + k * k;
    // The device part of the step: statistics [S | sum w] (weights resident, read in place), summed over the ranks once; closed
    // forms where the statistics lie; the Kronecker block; the conversion to free coordinates.  ONE stream: running what does
    // not need the statistics (the zeroing of the vector-coordinate matrix, the packing Jacobian, the sum of the weights) on a
    // side stream beside the pass was measured and lost -- 0.170 against 0.154 ms per step on the same box: the two event
    // hand-offs cost more than the seven short launches they take off the chain.
    auto chain = [&]() -> int {
        LRVB_TRY(free_conversion_clear(c, Vp));
        double* tiles = c->stats.p + 1 + c->P;
        if (!c->force_generic_wsyrk && q != 32 && q != 64) {
            // the Gram kernel leaves S as a dense q x q matrix and, through its spare column of ones, the sum of the weights beside it
            LRVB_TRY(launch_gram_small_on(c, c->X.p, c->N, q, c->w.p, tiles, c->qstats.p, q, c->qstats.p + q * q));
        } else {
            if (!c->force_generic_wsyrk) {
                LRVB_TRY(launch_gram_small_on(c, c->X.p, c->N, q, c->w.p, tiles, c->qstats.p, q));
            } else {
                LRVB_TRY(reserve_obs_vec(c, c->zbuf));
                HIP_TRY(hipMemcpyAsync(c->zbuf.p, c->w.p, (size_t)c->N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
                LRVB_TRY(launch_wsyrk(c, c->zbuf.p, tiles));
                LRVB_TRY(launch_tiles_to_dense(c, tiles, q, c->qstats.p, q, 0, 0, false));
            }
            hipLaunchKernelGGL(vec_block_sums_kernel, dim3(256), dim3(256), 0, c->stream, c->N, (const double*)c->w.p, c->qstats.p + q * q + 1);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)(c->qstats.p + q * q + 1), (i64)256, c->qstats.p + q * q);
            HIP_TRY(hipGetLastError());
        }
        LRVB_TRY(obs_reduce(c, c->qstats.p, q * q + 1));
        LRVB_TRY(launch_mvnreg_closed_forms(c, ix, c->qstats.p, hp_dev, scratch, c->g_eta.p, c->Heta.p, Gc, val));
        LRVB_TRY(launch_symkron3(c, (int)k, Gc, hp_dev + 32 + 2 * k, c->Heta.p, Vp, ix.ls));
        return free_conversion_padded(c, c->hprog.p, Vp);
    };
    // The chain is a dozen dependent launches of 3-15 us: from the THIRD call of one shape on it is replayed as a captured graph
    // (the first call warms every buffer, the second is captured).  Not under the profile marks, a sum-over-ranks hook or the
    // generic-kernel tuning bit, and never across a change of shape, stream or buffer addresses.
    lrvb_ctx::GraphSlot& gs = c->mv_graph;
    const i64 gkey[6] = { D, q, c->N, (i64)idx[0] | ((i64)idx[1] << 32), (i64)idx[2] | ((i64)idx[3] << 32), 0 };
    const bool graphable = !c->prof_on && !c->reduce_fn && !c->force_generic_wsyrk;
    const bool same = gs.epoch == c->buf_epoch && gs.stream == c->stream && memcmp(gs.key, gkey, sizeof(gkey)) == 0;
    if (graphable && same && gs.exec) {
        HIP_TRY(hipGraphLaunch(gs.exec, c->stream));
    } else if (graphable && same && gs.warmed) {
        // capture; if anything about it fails (a launch that cannot be captured, an instantiation error) nothing has run yet:
        // the chain is queued as plain launches and this slot stops trying until its key changes
        if (gs.exec) { (void)hipGraphExecDestroy(gs.exec); gs.exec = nullptr; }
        hipGraph_t graph = nullptr;
        bool ok = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            const int st = chain();
            const hipError_t ec = hipStreamEndCapture(c->stream, &graph);
            ok = st == LRVB_OK && ec == hipSuccess && graph != nullptr;
            if (ok) ok = hipGraphInstantiate(&gs.exec, graph, nullptr, nullptr, 0) == hipSuccess;
            if (graph) (void)hipGraphDestroy(graph);
        }
        if (ok) {
            HIP_TRY(hipGraphLaunch(gs.exec, c->stream));
        } else {
            (void)hipGetLastError();
            gs.exec = nullptr; gs.warmed = false; gs.broken = true;
            LRVB_TRY(chain());
        }
    } else {
        if (gs.exec) { (void)hipGraphExecDestroy(gs.exec); gs.exec = nullptr; }
        LRVB_TRY(chain());
        if (!same) gs.broken = false;                                    // (a failed capture is not retried for the same shape, stream and buffers)
        memcpy(gs.key, gkey, sizeof(gkey)); gs.epoch = c->buf_epoch; gs.stream = c->stream;
        gs.warmed = graphable && !gs.broken;
    }
    if (value_out) LRVB_TRY(d2h(c, value_out, val, 1));
    if (H_out) LRVB_TRY(d2h(c, H_out, c->Hfree.p, (size_t)D * (size_t)D));
    return LRVB_OK;
}

extern "C" int lrvb_lmm_global_hessian(lrvb_ctx* c, lrvb_ctx* gc, const double* free_val, int64_t n_free, const double* hp, int64_t n_hp,
                                       const int32_t* idx, double info_lb, double* sums_out, double* H_out) {
    LRVB_TRY(ctx_bind(c));
    LRVB_TRY(ctx_bind(gc));
    if (!free_val || !hp || !idx) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    if (c->device != gc->device) LRVB_FAIL(LRVB_ERR_INVALID, "the two contexts must live on one device");
    if (c->loss == LRVB_LOSS_NONE || !c->have_X) LRVB_FAIL(LRVB_ERR_STATE, "no data matrix: call lrvb_set_data(LRVB_SLOT_X) first");
    const i64 G = c->n_groups, q = c->P, p = q - 1, ng = gc->D, V = gc->V, Vp = V + (V & 1), mm = p * (p + 1) / 2;
    if (p < 1 || p + 7 > 64) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "1 <= p <= 57 regressors");
    if (V != ng || V != p + mm + 6) LRVB_FAIL(LRVB_ERR_INVALID, "the global context must hold the %lld global parameters of the model", (long long)(p + mm + 6));
    LRVB_TRY(check_len(n_free, ng + 2 * G, "free vector"));
    LRVB_TRY(check_len(n_hp, 32 + 2 * p + 3 * p * p, "host pack"));
    LmmIdx ix{ (int)p, idx[0], idx[1], idx[2], idx[3], idx[4], idx[5], idx[6], idx[7], Vp };
    const int32_t* ip = idx;
    if (ip[0] < 0 || ip[0] + p > V || ip[1] < 0 || ip[1] + mm > V) LRVB_FAIL(LRVB_ERR_INVALID, "parameter positions outside the global block");
    for (int t = 2; t < 8; ++t) if (ip[t] < 0 || ip[t] >= V) LRVB_FAIL(LRVB_ERR_INVALID, "parameter positions outside the global block");
    // every buffer first (an allocation may synchronise)
    LRVB_TRY(buf_reserve(gc, gc->hprog, (size_t)(ng + n_hp)));
    LRVB_TRY(buf_reserve(gc, gc->Heta, (size_t)Vp * (size_t)Vp));
    LRVB_TRY(buf_reserve(gc, gc->vtmp3, (size_t)(3 * p * p) > (size_t)V ? (size_t)(3 * p * p) : (size_t)V));
    LRVB_TRY(free_conversion_reserve(gc, Vp));
    // The whole step is ONE chain on the data context's stream: the global context's launches are queued there too (its own
    // stream waits once, before and after), so there is no event hand-off inside the step.
    // Uploads first: [par | local free parameters] for the elimination of the 2 G local parameters, [theta_g | hp] for the
    // closed forms.
    const hipStream_t gc_stream = gc->stream;
    LRVB_TRY(stream_handoff(gc, gc_stream, c->stream));        // whatever the global context still runs (a factorisation of the last result) comes first
    struct Restore { lrvb_ctx* g; hipStream_t st; ~Restore() { g->stream = st; } } restore{ gc, gc_stream };
    gc->stream = c->stream;
    std::vector<double> par((size_t)(8 + p));
    par[0] = hp[0]; par[1] = hp[1]; par[2] = hp[2]; par[3] = hp[8]; par[4] = hp[9]; par[5] = hp[10]; par[6] = hp[11]; par[7] = info_lb;
    memcpy(par.data() + 8, hp + 32, (size_t)p * sizeof(double));
    LmmTermsLayout L;
    LRVB_TRY(lmm_group_terms_prepare(c, par.data(), 8 + p, free_val + ng, 2 * G, L));
    std::vector<double> pack((size_t)(ng + n_hp));
    memcpy(pack.data(), free_val, (size_t)ng * sizeof(double));
    memcpy(pack.data() + ng, hp, (size_t)n_hp * sizeof(double));
    LRVB_TRY(h2d(gc, gc->hprog.p, pack.data(), pack.size()));
    const double* hp_dev = gc->hprog.p + ng;
    double* scratch = gc->vtmp3.p; double* Gc = scratch + 2 * p * p;
    double* sums = L.sums;
    auto chain = [&]() -> int {
        // (a) [S | group sums] in one pass, summed over the ranks; the 2 G local parameters eliminated
        LRVB_TRY(grouped_stats_device(c));
        LRVB_TRY(lmm_group_terms_launch(c, L, false));
        // (b) the closed forms where the statistics lie (the sums over groups formed inside the kernel from the partial rows),
        // the Kronecker block, the conversion to free coordinates
        LRVB_TRY(free_conversion_clear(gc, Vp));
        LRVB_TRY(launch_lmm_closed_forms(gc, ix, c->gstats.p, sums, sums + 128, hp_dev, scratch, gc->g_eta.p, gc->Heta.p, Gc, L.part, (int)L.n_waves));
        LRVB_TRY(launch_symkron3(gc, (int)p, Gc, hp_dev + 32 + 2 * p, gc->Heta.p, Vp, ix.ls));
        return free_conversion_padded(gc, gc->hprog.p, Vp);
    };
    // (Replaying this chain as a captured graph, as lrvb_mvnreg_hessian does, was built and measured: 296 against 291 us per
    // step.  The host queues these ~19 launches in a third of the time the device needs for them -- the 90 us statistics kernel
    // goes first -- so the device never waits for a launch; configuration 2's chain of 5 us kernels does.)
    LRVB_TRY(chain());
    gc->stream = gc_stream;
    LRVB_TRY(stream_handoff(c, c->stream, gc_stream));          // the global context's own stream (lrvb_chol_factor_last, copies) continues behind the step
    if (sums_out) LRVB_TRY(d2h(c, sums_out, sums, 128));
    if (H_out) LRVB_TRY(d2h(gc, H_out, gc->Hfree.p, (size_t)ng * (size_t)ng));
    return LRVB_OK;
}

// ---- cross Hessians ----------------------------------------------------------------------

static int obs_grad_impl(lrvb_ctx* c, const double* point, i64 n_in, bool is_free, i64 n0, i64 n1, double* G_out) {
    LRVB_TRY(ctx_bind(c));
    if (!point || !G_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    const i64 width = is_free ? c->D : c->V;
    LRVB_TRY(check_len(n_in, width, is_free ? "free vector" : "vector"));
    if (c->loss == LRVB_LOSS_NONE) LRVB_FAIL(LRVB_ERR_STATE, "model has no data term");
    if (n0 < 0 || n1 > c->N || n0 > n1) LRVB_FAIL(LRVB_ERR_INVALID, "row range [%lld, %lld) outside [0, %lld)", (long long)n0, (long long)n1, (long long)c->N);
    LRVB_TRY(data_ready(c));
    LRVB_TRY(h2d(c, c->theta.p, point, (size_t)width));
    LRVB_TRY(set_point(c, c->theta.p, is_free));
    LRVB_TRY(eval_grad_eta(c, c->stats.p, false, false));        // per-observation l' only: rank-local
    const bool diag_path = !is_free || c->all_box;
    if (!is_free) EW(fill_kernel, c->V, 1.0, c->vtmp.p);
    if (is_free && !c->all_box) LRVB_TRY(ensure_dense_J(c, c->theta.p));
    const i64 chunk = 16384;
    for (i64 a = n0; a < n1; a += chunk) {
        const i64 b = (a + chunk < n1) ? a + chunk : n1;
        const i64 rows = b - a;
        LRVB_TRY(buf_reserve(c, c->rhs, (size_t)rows * (size_t)width));
        if (diag_path) {
            LRVB_TRY(launch_obs_grad(c, a, b, c->rhs.p, 0, is_free ? c->j1.p : c->vtmp.p));
        } else {
            LRVB_TRY(buf_reserve(c, c->work1, (size_t)rows * (size_t)c->P));
            LRVB_TRY(launch_obs_grad(c, a, b, c->work1.p, 1, nullptr));
            LRVB_TRY(launch_gemm(c, false, false, rows, c->D, c->P, 1.0, c->work1.p, c->P,
                                 c->Jdense.p + c->glm_off * c->D, c->D, 0.0, c->rhs.p, c->D));
        }
        LRVB_TRY(d2h(c, G_out + (a - n0) * width, c->rhs.p, (size_t)rows * (size_t)width));
    }
    return LRVB_OK;
}
extern "C" int lrvb_obs_grad(lrvb_ctx* c, const double* free_in, int64_t D, int64_t n0, int64_t n1, double* G_out) {
    return obs_grad_impl(c, free_in, D, true, n0, n1, G_out);
}
extern "C" int lrvb_obs_grad_vec(lrvb_ctx* c, const double* vec_in, int64_t V, int64_t n0, int64_t n1, double* G_out) {
    return obs_grad_impl(c, vec_in, V, false, n0, n1, G_out);
}

extern "C" int lrvb_cross_hessian_tilt(lrvb_ctx* c, const double* free_in, int64_t D, double* C_out) {
    LRVB_TRY(ctx_bind(c));
    if (!free_in || !C_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(check_len(D, c->D, "free vector"));
    if (c->quad_kind == LRVB_QUAD_NONE) LRVB_FAIL(LRVB_ERR_STATE, "model has no quadratic term");
    LRVB_TRY(h2d(c, c->theta.p, free_in, (size_t)D));
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)c->V * (size_t)c->D));
    if (c->all_box) {
        LRVB_TRY(set_point(c, c->theta.p, true));
        dim3 grid(nb256(c->V), (unsigned)c->D);
        hipLaunchKernelGGL(diag_scale_kernel, grid, dim3(256), 0, c->stream, c->D, c->V, c->quad_scale, c->j1.p, c->work1.p);
        HIP_TRY(hipGetLastError());
    } else {
        LRVB_TRY(ensure_dense_J(c, c->theta.p));
        LRVB_TRY(buf_reserve(c, c->Heta, (size_t)c->V * (size_t)c->D));
        dim3 grid(nb256(c->D), (unsigned)c->V);
        hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, c->stream, c->V, c->D, c->Jdense.p, c->work1.p);
        HIP_TRY(hipGetLastError());
        LRVB_TRY(launch_axpby(c, c->V * c->D, c->quad_scale, c->work1.p, 0.0, c->work1.p));
    }
    return d2h(c, C_out, c->work1.p, (size_t)c->D * (size_t)c->V);
}

// ---- the other hyper-parameters of the declared objective (kernels: k_hyper.hip) -----------------------------------
extern "C" int lrvb_hyper_size(lrvb_ctx* c, int kind, int64_t* n_hyper) {
    if (!c || !n_hyper) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    if (kind == LRVB_HYPER_LIK_INFO) {
        if (c->loss != LRVB_LOSS_GAUSSIAN) LRVB_FAIL(LRVB_ERR_STATE, "lik_info is the precision of the Gaussian loss");
        *n_hyper = 1; return LRVB_OK;
    }
    if (kind < LRVB_HYPER_TILT || kind > LRVB_HYPER_LIK_INFO) LRVB_FAIL(LRVB_ERR_INVALID, "unknown hyper-parameter kind %d", kind);
    if (c->quad_kind == LRVB_QUAD_NONE) LRVB_FAIL(LRVB_ERR_STATE, "model has no quadratic term");
    if (kind == LRVB_HYPER_QUAD_SCALE) *n_hyper = 1;
    else if (kind == LRVB_HYPER_QUAD_A && c->quad_kind == LRVB_QUAD_DENSE) *n_hyper = c->V * (c->V + 1) / 2;
    else *n_hyper = c->V;
    return LRVB_OK;
}
// theta / eta (and j1, the dense J of general layouts) at the point; r = eta - m in vtmp, A r in vtmp2 for the quadratic kinds;
// the data gradient in g_eta and the data value in stats[0] for lik_info
static int hyper_point(lrvb_ctx* c, int kind, const double* point, i64 n_in, bool is_free, int64_t n_hyper, bool need_J) {
    LRVB_TRY(ctx_bind(c));
    if (!point) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    int64_t Ph = 0;
    LRVB_TRY(lrvb_hyper_size(c, kind, &Ph));
    LRVB_TRY(check_len(n_hyper, Ph, "hyper-parameter"));
    const i64 n = is_free ? c->D : c->V;
    LRVB_TRY(check_len(n_in, n, is_free ? "free vector" : "vector"));
    LRVB_TRY(h2d(c, c->theta.p, point, (size_t)n));
    LRVB_TRY(set_point(c, c->theta.p, is_free));
    if (need_J && is_free && !c->all_box) LRVB_TRY(ensure_dense_J(c, c->theta.p));
    if (kind == LRVB_HYPER_LIK_INFO) {
        LRVB_TRY(data_ready(c));
        LRVB_TRY(eval_grad_eta(c, c->stats.p, false, true));
    } else {
        LRVB_TRY(launch_quad_diff(c, c->eta.p));
    }
    return LRVB_OK;
}
extern "C" int lrvb_cross_hessian_hyper(lrvb_ctx* c, int kind, const double* point, int64_t n_in, int is_free,
                                        double* C_out, int64_t n_hyper) {
    if (!C_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(hyper_point(c, kind, point, n_in, is_free != 0, n_hyper, true));
    const i64 V = c->V, D = c->D, Ph = n_hyper, n = is_free ? D : V;
    if ((double)V * (double)Ph > 2.0e9) LRVB_FAIL(LRVB_ERR_SIZE, "a %lld x %lld cross Hessian does not fit: take a sub-block of the hyper-parameter", (long long)V, (long long)Ph);
    const double* col = nullptr;
    if (kind == LRVB_HYPER_QUAD_SCALE) {                      // d/ds of s (A r + b)
        LRVB_TRY(launch_hyper_col(c, 1.0, c->vtmp2.p, 1.0, c->quadB.p, c->vtmp3.p));
        col = c->vtmp3.p;
    } else if (kind == LRVB_HYPER_LIK_INFO) {                 // the data gradient is linear in tau
        LRVB_TRY(launch_hyper_col(c, 1.0 / c->lik_info, c->g_eta.p, 0.0, nullptr, c->vtmp3.p));
        col = c->vtmp3.p;
    }
    const bool diag_J = is_free && c->all_box;
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)V * (size_t)Ph));
    LRVB_TRY(launch_hyper_cross(c, kind, Ph, c->vtmp.p, col, diag_J ? c->j1.p : nullptr, c->work1.p));
    if (!is_free || diag_J) return d2h(c, C_out, c->work1.p, (size_t)n * (size_t)Ph);
    LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)D * (size_t)Ph));
    LRVB_TRY(launch_gemm(c, true, false, D, Ph, V, 1.0, c->Jdense.p, D, c->work1.p, Ph, 0.0, c->Hfree.p, Ph));
    return d2h(c, C_out, c->Hfree.p, (size_t)D * (size_t)Ph);
}
extern "C" int lrvb_hyper_grad(lrvb_ctx* c, int kind, const double* point, int64_t n_in, int is_free,
                               double* g_out, int64_t n_hyper) {
    if (!g_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(hyper_point(c, kind, point, n_in, is_free != 0, n_hyper, false));
    if (kind == LRVB_HYPER_LIK_INFO) {                        // sum_n w_n l_n is linear in tau
        double v = 0.0;
        LRVB_TRY(d2h(c, &v, c->stats.p, 1));
        g_out[0] = v / c->lik_info;
        return LRVB_OK;
    }
    if (kind == LRVB_HYPER_QUAD_SCALE) {                      // 1/2 r^T A r + b^T eta: the quadratic term at unit scale
        const double saved = c->quad_scale;
        c->quad_scale = 1.0;
        hipError_t e = hipMemsetAsync(c->scal.p, 0, sizeof(double), c->stream);
        int st = (e == hipSuccess) ? launch_quad_grad_value(c, c->eta.p, nullptr, c->scal.p) : LRVB_ERR_HIP;
        c->quad_scale = saved;
        if (e != hipSuccess) LRVB_FAIL(LRVB_ERR_HIP, "memset failed: %s", hipGetErrorString(e));
        LRVB_TRY(st);
        return d2h(c, g_out, c->scal.p, 1);
    }
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)n_hyper));
    LRVB_TRY(launch_hyper_grad(c, kind, n_hyper, c->eta.p, c->vtmp.p, c->vtmp2.p, c->work1.p));
    return d2h(c, g_out, c->work1.p, (size_t)n_hyper);
}
extern "C" int lrvb_jac_t_matmul(lrvb_ctx* c, const double* free_in, int64_t D, const double* B, int64_t Q, double* out) {
    LRVB_TRY(ctx_bind(c));
    if (!free_in || !B || !out || Q <= 0) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    LRVB_TRY(check_len(D, c->D, "free vector"));
    const i64 V = c->V;
    LRVB_TRY(h2d(c, c->theta.p, free_in, (size_t)D));
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)V * (size_t)Q));
    LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)D * (size_t)Q));
    LRVB_TRY(h2d(c, c->work1.p, B, (size_t)V * (size_t)Q));
    if (c->jt_rows > 0) {
        LRVB_TRY(launch_jt_apply(c, c->theta.p, c->work1.p, Q, Q, c->Hfree.p, Q, false));      // the structured product (k_pack.hip)
    } else {
        LRVB_TRY(ensure_dense_J(c, c->theta.p));
        LRVB_TRY(launch_gemm(c, true, false, D, Q, V, 1.0, c->Jdense.p, D, c->work1.p, Q, 0.0, c->Hfree.p, Q));
    }
    return d2h(c, out, c->Hfree.p, (size_t)D * (size_t)Q);
}

static int gram_dev_impl(lrvb_ctx* c, const double* free_dev, double* G_dev, i64 ld) {
    if (c->loss == LRVB_LOSS_NONE) LRVB_FAIL(LRVB_ERR_STATE, "model has no data term");
    if (ld < c->D) LRVB_FAIL(LRVB_ERR_SIZE, "leading dimension too small");
    LRVB_TRY(data_ready(c));
    LRVB_TRY(set_point(c, free_dev, true));
    LRVB_TRY(eval_grad_eta(c, c->stats.p, false, false));
    LRVB_TRY(reserve_obs_vec(c, c->zbuf));
    EW(square_kernel, c->N, c->lp.p, c->zbuf.p);
    double* tiles = c->stats.p + 1 + c->P;
    LRVB_TRY(launch_wsyrk(c, c->zbuf.p, tiles));
    LRVB_TRY(obs_reduce(c, tiles, (i64)wsyrk_num_tiles(c->P) * WS_TILE * WS_TILE));
    const int saved_quad = c->quad_kind;
    c->quad_kind = LRVB_QUAD_NONE;               // G^T G has no quadratic-term contribution
    int st;
    if (c->all_box) {
        st = launch_finish_box(c, tiles, nullptr, c->j1.p, nullptr, false, G_dev, ld);
    } else {
        st = buf_reserve(c, c->Heta, (size_t)c->V * (size_t)c->V);
        if (st == LRVB_OK) st = buf_reserve(c, c->work1, (size_t)c->V * (size_t)c->D);
        if (st == LRVB_OK) st = launch_build_Heta(c, tiles, c->Heta.p);
        if (c->jt_rows > 0) {                                    // the structured products (k_pack.hip)
            if (st == LRVB_OK) st = launch_jt_apply(c, free_dev, c->Heta.p, c->V, c->V, c->work1.p, c->V, false);
            if (st == LRVB_OK) st = launch_jt_apply(c, free_dev, c->work1.p, c->V, c->D, G_dev, ld, true);
        } else {
            if (st == LRVB_OK) st = ensure_dense_J(c, free_dev);
            if (st == LRVB_OK) st = launch_gemm(c, false, false, c->V, c->D, c->V, 1.0, c->Heta.p, c->V, c->Jdense.p, c->D, 0.0, c->work1.p, c->D);
            if (st == LRVB_OK) st = launch_gemm(c, true, false, c->D, c->D, c->V, 1.0, c->Jdense.p, c->D, c->work1.p, c->D, 0.0, G_dev, ld);
        }
    }
    c->quad_kind = saved_quad;
    return st;
}
extern "C" int lrvb_gram_dev(lrvb_ctx* c, const double* free_dev, double* GtG_dev, int64_t ld) {
    LRVB_TRY(ctx_bind(c));
    if (!free_dev || !GtG_dev) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    return gram_dev_impl(c, free_dev, GtG_dev, ld);
}
extern "C" int lrvb_gram(lrvb_ctx* c, const double* free_in, int64_t D, double* GtG_out, int64_t ld) {
    LRVB_TRY(ctx_bind(c));
    if (!free_in || !GtG_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(check_len(D, c->D, "free vector"));
    if (ld < D) LRVB_FAIL(LRVB_ERR_SIZE, "leading dimension too small");
    LRVB_TRY(h2d(c, c->theta.p, free_in, (size_t)D));
    LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)D * (size_t)D));
    LRVB_TRY(gram_dev_impl(c, c->theta.p, c->Hfree.p, D));
    HIP_TRY(hipMemcpy2DAsync(GtG_out, (size_t)ld * 8, c->Hfree.p, (size_t)D * 8, (size_t)D * 8, (size_t)D, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LRVB_OK;
}

// ---- objectives quadratic in the data -----------------------------------------------------------


static int weighted_gram_impl(lrvb_ctx* c, double* S_out, int64_t ld, double* wsum_out);
extern "C" int lrvb_weighted_gram(lrvb_ctx* c, double* S_out, int64_t ld) { return weighted_gram_impl(c, S_out, ld, nullptr); }
extern "C" int lrvb_weighted_gram_sum(lrvb_ctx* c, double* S_out, int64_t ld, double* wsum_out) {
    if (!wsum_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    return weighted_gram_impl(c, S_out, ld, wsum_out);
}
static int weighted_gram_impl(lrvb_ctx* c, double* S_out, int64_t ld, double* wsum_out) {
    LRVB_TRY(ctx_bind(c));
    if (!S_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    if (c->loss == LRVB_LOSS_NONE || !c->have_X) LRVB_FAIL(LRVB_ERR_STATE, "no data matrix: call lrvb_set_data(LRVB_SLOT_X) first");
    if (ld < c->P) LRVB_FAIL(LRVB_ERR_SIZE, "leading dimension too small");
    double* tiles = c->stats.p + 1 + c->P;
    if (c->P <= 64 && !c->force_generic_wsyrk) {
        LRVB_TRY(launch_gram_small_on(c, c->X.p, c->N, c->P, c->w.p, tiles));   // reads the resident weights in place (no padding needed)
    } else {                                                     // c = w (padded copy: the LDS-DMA stage over-reads up to 31 entries past N)
        LRVB_TRY(reserve_obs_vec(c, c->zbuf));
        HIP_TRY(hipMemcpyAsync(c->zbuf.p, c->w.p, (size_t)c->N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        LRVB_TRY(launch_wsyrk(c, c->zbuf.p, tiles));
    }
    const i64 PP = c->P * c->P;
    LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)PP + 1 + 256));
    LRVB_TRY(launch_tiles_to_dense(c, tiles, c->P, c->Hfree.p, c->P, 0, 0, false));
    if (wsum_out) {                                              // sum of the weights rides in the same buffer: [S | sum w]
        hipLaunchKernelGGL(vec_block_sums_kernel, dim3(256), dim3(256), 0, c->stream, c->N, (const double*)c->w.p, c->Hfree.p + PP + 1);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)(c->Hfree.p + PP + 1), (i64)256, c->Hfree.p + PP);
        HIP_TRY(hipGetLastError());
    }
    LRVB_TRY(obs_reduce(c, c->Hfree.p, PP + (wsum_out ? 1 : 0)));          // observation shards: summed on the device, ONCE, before the copy
    HIP_TRY(hipMemcpy2DAsync(S_out, (size_t)ld * 8, c->Hfree.p, (size_t)c->P * 8, (size_t)c->P * 8, (size_t)c->P, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (wsum_out) LRVB_TRY(d2h(c, wsum_out, c->Hfree.p + PP, 1));
    return LRVB_OK;
}

extern "C" int lrvb_obs_quadform(lrvb_ctx* c, const double* M, const double* cvec, int64_t K,
                                 int64_t n0, int64_t n1, double* out) {
    LRVB_TRY(ctx_bind(c));
    if (!M || !out || K <= 0) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    if (c->loss == LRVB_LOSS_NONE || !c->have_X) LRVB_FAIL(LRVB_ERR_STATE, "no data matrix: call lrvb_set_data(LRVB_SLOT_X) first");
    if (n0 < 0 || n1 > c->N || n0 > n1) LRVB_FAIL(LRVB_ERR_INVALID, "row range [%lld, %lld) outside [0, %lld)", (long long)n0, (long long)n1, (long long)c->N);
    const size_t msz = (size_t)K * (size_t)c->P * (size_t)c->P;
    LRVB_TRY(buf_reserve(c, c->work1, msz + (size_t)K));
    LRVB_TRY(h2d(c, c->work1.p, M, msz));
    double* cdev = nullptr;
    if (cvec) { cdev = c->work1.p + msz; LRVB_TRY(h2d(c, cdev, cvec, (size_t)K)); }
    const i64 chunk = 32768;
    for (i64 a = n0; a < n1; a += chunk) {
        const i64 b = (a + chunk < n1) ? a + chunk : n1;
        const i64 rows = b - a;
        LRVB_TRY(buf_reserve(c, c->rhs, (size_t)rows * (size_t)K));
        dim3 grid(nb256(K), (unsigned)rows);
        hipLaunchKernelGGL(obs_quadform_kernel, grid, dim3(256), 0, c->stream, c->X.p, c->P, (int)c->P,
                           c->work1.p, cdev, (i64)K, a, b, c->rhs.p);
        HIP_TRY(hipGetLastError());
        LRVB_TRY(d2h(c, out + (a - n0) * K, c->rhs.p, (size_t)rows * (size_t)K));
    }
    return LRVB_OK;
}

// ---- grouped sufficient statistics (hierarchical models: BASELINE.json config 4) -----------------

extern "C" int lrvb_set_groups(lrvb_ctx* c, const int32_t* gid, int64_t n, int64_t n_groups) {
    LRVB_TRY(ctx_bind(c));
    if (!gid || n != c->N || n_groups <= 0) LRVB_FAIL(LRVB_ERR_SIZE, "group ids must have %lld entries and n_groups > 0", (long long)c->N);
    if (c->P > 64) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "grouped sums support n_cols <= 64");
    std::vector<i64> offs((size_t)n_groups + 1, 0), perm((size_t)n);
    for (i64 i = 0; i < n; ++i) {
        if (gid[i] < 0 || gid[i] >= n_groups) LRVB_FAIL(LRVB_ERR_INVALID, "group id %d of row %lld outside [0, %lld)", gid[i], (long long)i, (long long)n_groups);
        offs[(size_t)gid[i] + 1]++;
    }
    for (i64 g = 0; g < n_groups; ++g) offs[(size_t)g + 1] += offs[(size_t)g];
    std::vector<i64> cur(offs.begin(), offs.end() - 1);
    for (i64 i = 0; i < n; ++i) perm[(size_t)cur[(size_t)gid[i]]++] = i;      // stable: rows of a group keep their order
    // the fused one-pass statistics kernel (k_lmm.hip) gives wave w the rows [w R, (w + 1) R) of the sorted order:
    // the group that holds each wave's first row
    const i64 R = grouped_rows_per_wave(n), NW = (n + R - 1) / R;
    std::vector<i64> wg0((size_t)NW);
    for (i64 wv = 0; wv < NW; ++wv)
        wg0[(size_t)wv] = (i64)(std::upper_bound(offs.begin(), offs.end(), wv * R) - offs.begin()) - 1;
    const size_t words = (size_t)n + (size_t)n_groups + 1 + (size_t)NW;
    LRVB_TRY(buf_reserve(c, c->groups, words));          // i64 and double are both 8 bytes
    i64* dev = reinterpret_cast<i64*>(c->groups.p);
    HIP_TRY(hipMemcpyAsync(dev, perm.data(), (size_t)n * sizeof(i64), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dev + n, offs.data(), ((size_t)n_groups + 1) * sizeof(i64), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(dev + n + n_groups + 1, wg0.data(), (size_t)NW * sizeof(i64), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->n_groups = n_groups;
    c->gstats_valid = false;
    c->zs_valid = false; c->ws_valid = false;
    return LRVB_OK;
}

extern "C" int lrvb_group_sums(lrvb_ctx* c, double* out) {
    LRVB_TRY(ctx_bind(c));
    if (!out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    if (c->n_groups <= 0) LRVB_FAIL(LRVB_ERR_STATE, "no groups: call lrvb_set_groups first");
    if (!c->have_X) LRVB_FAIL(LRVB_ERR_STATE, "no data matrix");
    const i64 G = c->n_groups;
    const size_t n_out = (size_t)G * (size_t)(c->P + 1);
    LRVB_TRY(buf_reserve(c, c->work1, n_out));
    const i64* dev = reinterpret_cast<const i64*>(c->groups.p);
    hipLaunchKernelGGL(group_sums_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, c->stream,
                       c->X.p, c->P, (int)c->P, c->w.p, dev, dev + c->N, G, c->work1.p);
    HIP_TRY(hipGetLastError());
    LRVB_TRY(obs_reduce(c, c->work1.p, (i64)n_out));
    return d2h(c, out, c->work1.p, n_out);
}

// ---- hierarchical LMM (config 4): statistics resident on the device, group effects eliminated there ---------------
// [S (q x q) | group sums (G x (q + 1))] in ONE device buffer, handed to the sum-over-ranks hook once and kept
// resident for lrvb_lmm_group_terms; both host copies are optional.
static int grouped_stats_device(lrvb_ctx* c);
extern "C" int lrvb_grouped_stats(lrvb_ctx* c, double* S_out, double* gs_out) {
    LRVB_TRY(ctx_bind(c));
    LRVB_TRY(grouped_stats_device(c));
    const i64 G = c->n_groups, q = c->P;
    const size_t n_s = (size_t)q * q, n_g = (size_t)G * (size_t)(q + 1);
    if (S_out) LRVB_TRY(d2h(c, S_out, c->gstats.p, n_s));
    if (gs_out) LRVB_TRY(d2h(c, gs_out, c->gstats.p + n_s, n_g));
    return LRVB_OK;
}
// [S | group sums] of the context's rows and weights into c->gstats, summed over the ranks: no host copy
static int grouped_stats_device(lrvb_ctx* c) {
    if (c->n_groups <= 0) LRVB_FAIL(LRVB_ERR_STATE, "no groups: call lrvb_set_groups first");
    if (c->loss == LRVB_LOSS_NONE || !c->have_X) LRVB_FAIL(LRVB_ERR_STATE, "no data matrix: call lrvb_set_data(LRVB_SLOT_X) first");
    const i64 G = c->n_groups, q = c->P;
    const size_t n_s = (size_t)q * q, n_g = (size_t)G * (size_t)(q + 1);
    LRVB_TRY(buf_reserve(c, c->gstats, n_s + n_g));
    c->gstats_valid = false;
    if (grouped_fused_supported(c) && !c->force_generic_wsyrk) {
        // one pass over the group-sorted rows: Gram on the matrix cores and the per-group sums from the same registers
        LRVB_TRY(launch_grouped_stats_fused(c, c->gstats.p, c->gstats.p + n_s));
    } else {
        // odd row length (8-byte rows cannot take 16-byte loads): the narrow Gram kernel, then one wavefront per group
        LRVB_TRY(reserve_obs_vec(c, c->zbuf));
        HIP_TRY(hipMemcpyAsync(c->zbuf.p, c->w.p, (size_t)c->N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        double* tiles = c->stats.p + 1 + c->P;
        LRVB_TRY(launch_wsyrk(c, c->zbuf.p, tiles));
        LRVB_TRY(launch_tiles_to_dense(c, tiles, q, c->gstats.p, q, 0, 0, false));
        const i64* dev = reinterpret_cast<const i64*>(c->groups.p);
        if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
        hipLaunchKernelGGL(group_sums_kernel, dim3((unsigned)((G + 3) / 4)), dim3(256), 0, c->stream,
                           c->X.p, c->P, (int)c->P, c->w.p, dev, dev + c->N, G, c->gstats.p + n_s);
        HIP_TRY(hipGetLastError());
        if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    }
    LRVB_TRY(obs_reduce(c, c->gstats.p, (i64)(n_s + n_g)));
    c->gstats_valid = true;
    return LRVB_OK;
}


// par (host, 8 + p): [ty, tm, e_mu, d ty / d a_y, d ty / d b_y, d tm / d a_mu, d tm / d b_mu, lower bound of the local
// informations, m (p)]; f_local (host, 2 G): the FREE local parameters [e_1..e_G | log(i_g - lb)].  out (host,
// 128 + (p + 5)^2): the sums of lmm_group_kernel, then M = sum_g c_e c_e^T / dfe + c_i c_i^T / dfi -- the Schur
// complement of the 2 G local parameters onto the coupled global rows, in vector coordinates of the globals and free
// coordinates of the locals (H_gl diag(H_ll)^-1 H_lg; the G independent 2 x 2 local blocks of this model are diagonal).
static int lmm_group_terms_device(lrvb_ctx* c, const double* par, int64_t n_par, const double* f_local, int64_t n_local, double** sums_dev);
extern "C" int lrvb_lmm_group_terms(lrvb_ctx* c, const double* par, int64_t n_par, const double* f_local, int64_t n_local, double* out) {
    LRVB_TRY(ctx_bind(c));
    if (!par || !f_local || !out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    double* sums = nullptr;
    LRVB_TRY(lmm_group_terms_device(c, par, n_par, f_local, n_local, &sums));
    const i64 R = c->P - 1 + 5;
    // [sums (128) | M (R x R)] are adjacent: one copy
    return d2h(c, out, sums, (size_t)(128 + R * R));
}
// the same, the result [sums (128) | M (R x R)] left on the device (in c->work1; valid until the next use of that buffer).
// Two halves, so that lrvb_lmm_global_hessian can put every upload of a step in front of its launch chain (and replay the chain as
// a captured graph): `prepare` checks, reserves and uploads [par | f_local]; `launch` queues the kernels.
static int lmm_group_terms_prepare(lrvb_ctx* c, const double* par, int64_t n_par, const double* f_local, int64_t n_local, LmmTermsLayout& L) {
    const i64 G = c->n_groups, q = c->P, p = q - 1, R = p + 5;
    if (G <= 0) LRVB_FAIL(LRVB_ERR_STATE, "no groups: call lrvb_set_groups first");
    if (p < 1 || p + 7 > 64) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "1 <= p <= 57 regressors");
    LRVB_TRY(check_len(n_par, 8 + p, "par"));
    LRVB_TRY(check_len(n_local, 2 * G, "local free vector"));
    const int ldc = (int)((R + 1) & ~(i64)1);                     // even width: 16-byte loads in the narrow Gram kernel
    i64 grid = (G + 15) / 16; if (grid > 1024) grid = 1024; if (grid < 1) grid = 1;     // four groups per wave; one partial row per workgroup
    const i64 n_waves = grid;
    const size_t nC = (size_t)(2 * G + 16) * (size_t)ldc, nW = (size_t)(2 * G + 64);
    LRVB_TRY(buf_reserve(c, c->work1, nC + nW + (size_t)(8 + p) + (size_t)(2 * G) + (size_t)n_waves * 128 + 128 + 64 * 64));
    LRVB_TRY(buf_reserve(c, c->Tdense, (size_t)WS_TILE * WS_TILE));
    L.Cm = c->work1.p; L.wts = L.Cm + nC; L.dpar = L.wts + nW; L.dloc = L.dpar + (8 + p);
    L.part = L.dloc + 2 * G; L.sums = L.part + n_waves * 128; L.Md = L.sums + 128;
    L.ldc = ldc; L.grid = grid; L.n_waves = n_waves; L.G = G; L.p = p; L.R = R;
    // dpar and dloc are adjacent: one upload
    std::vector<double> pack((size_t)(8 + p + 2 * G));
    memcpy(pack.data(), par, (size_t)(8 + p) * sizeof(double));
    memcpy(pack.data() + 8 + p, f_local, (size_t)(2 * G) * sizeof(double));
    return h2d(c, L.dpar, pack.data(), pack.size());
}
// with_sums = false: the sums over groups are left to the consumer of the partial rows (lmm_closed_forms_kernel forms them itself)
static int lmm_group_terms_launch(lrvb_ctx* c, const LmmTermsLayout& L, bool with_sums) {
    if (!c->gstats_valid) LRVB_FAIL(LRVB_ERR_STATE, "no grouped statistics resident: call lrvb_grouped_stats first");
    const i64 q = c->P;
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    hipLaunchKernelGGL(lmm_group_kernel, dim3((unsigned)L.grid), dim3(256), 0, c->stream,
                       (const double*)(c->gstats.p + q * q), L.G, (int)L.p, (const double*)L.dpar, (const double*)L.dloc, L.Cm, L.ldc, L.wts, L.part);
    HIP_TRY(hipGetLastError());
    if (with_sums) {
        hipLaunchKernelGGL(lmm_sums_kernel, dim3(1), dim3(1024), 0, c->stream, (const double*)L.part, (int)L.n_waves, L.sums);
        HIP_TRY(hipGetLastError());
    }
    if (c->prof_on) LRVB_TRY(prof_mark(c, PROF_WSYRK));
    // M as a dense R x R matrix, no unpacking launch (the Gram kernel clamps rows past 2 G: no padding to clear)
    return launch_gram_small_on(c, L.Cm, 2 * L.G, L.ldc, L.wts, c->Tdense.p, L.Md, L.R, nullptr, L.R);
}
static int lmm_group_terms_device(lrvb_ctx* c, const double* par, int64_t n_par, const double* f_local, int64_t n_local, double** sums_dev) {
    if (!c->gstats_valid) LRVB_FAIL(LRVB_ERR_STATE, "no grouped statistics resident: call lrvb_grouped_stats first");
    LmmTermsLayout L;
    LRVB_TRY(lmm_group_terms_prepare(c, par, n_par, f_local, n_local, L));
    LRVB_TRY(lmm_group_terms_launch(c, L, true));
    *sums_dev = L.sums;
    return LRVB_OK;
}

// ---- mixture model: per-row simplex blocks eliminated on the device (config 3) --------------------
// Inputs: free local parameters theta_z (N x (K-1)), Lam ((V+1) x K) = [E log pi; E log phi].
// Outputs: val2 = [-sum w z.s, sum w z log z], the free local gradient (N x (K-1)), the weighted
// sufficient statistics S64 = U^T diag(w) U with U = [x~ (32) | z (32)], and
// R ((V+1)^2 x K^2) = sum_n w_n^2 (x~_n (x) x~_n) vec(J_n H_nn^-1 J_n^T)^T  (the Schur-complement term).
// mx_flags bit 0: form the Schur operand and leave it on the device (no copy to R_out, which may be NULL)
static int mixture_rows_impl(lrvb_ctx* c, int32_t K, const double* theta_z, const double* Lam,
                             double* val2_out, double* gfree_out, double* S64_out, double* R_out, int mx_flags) {
    LRVB_TRY(ctx_bind(c));
    if (!Lam || !val2_out || !S64_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    if (c->loss == LRVB_LOSS_NONE || !c->have_X) LRVB_FAIL(LRVB_ERR_STATE, "no data matrix: call lrvb_set_data(LRVB_SLOT_X) first");
    const i64 N = c->N;
    if (!theta_z && c->mx_theta_n != N * (i64)(K - 1))
        LRVB_FAIL(LRVB_ERR_STATE, "theta_z is NULL and no simplex logits of this shape are resident from an earlier call");
    const int V = (int)c->P;
    if (V + 1 > 32 || K > 32 || K < 2) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "mixture kernel supports V + 1 <= 32 and 2 <= K <= 32");
    const i64 KM = K - 1, KK = (i64)K * K, QQ = (i64)(V + 1) * (V + 1);
    // both factors of the Schur operand are symmetric matrices: packed lower triangles
    const i64 KP = (i64)K * (K + 1) / 2, QP = (i64)(V + 1) * (V + 2) / 2;
    const i64 lda = KP + (KP & 1), ldk = QP + (QP & 1);
    DevBuf &thz = c->mx_theta, &lam = c->mx_lam, &Amat = c->mx_A, &U = c->mx_U, &gfr = c->mx_g, &Xk = c->mx_Xk, &Rd = c->mx_R;
    c->x2_ready = false;
    c->mx_res_K = c->mx_res_q = 0;
    int st = buf_reserve(c, thz, (size_t)(N * KM));
    if (st == LRVB_OK) st = buf_reserve(c, lam, (size_t)((V + 1) * K));
    if (st == LRVB_OK) st = buf_reserve(c, Amat, (size_t)((N + 16) * lda));      // + 16 rows of zeros: the sliver loads of launch_atb run past N
    if (st == LRVB_OK) st = buf_reserve(c, U, (size_t)(N * 64));
    if (st == LRVB_OK && gfree_out) st = buf_reserve(c, gfr, (size_t)(N * KM));
    if (st == LRVB_OK && theta_z) { c->mx_theta_n = 0; st = h2d(c, thz.p, theta_z, (size_t)(N * KM)); if (st == LRVB_OK) c->mx_theta_n = N * KM; }
    if (st == LRVB_OK) st = h2d(c, lam.p, Lam, (size_t)((V + 1) * K));
    int* bad = reinterpret_cast<int*>(c->scal.p + 8);
    if (st == LRVB_OK && c->prof_on) st = prof_mark(c, PROF_WSYRK);                     // the per-row kernel counts among the statistics kernels
    // (the local gradient is 8 N (K - 1) bytes of writes: only formed when the caller takes it)
    if (st == LRVB_OK) st = launch_mixture_rows(c, K, thz.p, lam.p, Amat.p, lda, U.p, gfree_out ? gfr.p : (double*)nullptr, c->scal.p, bad);
    if (st == LRVB_OK && c->prof_on) st = prof_mark(c, PROF_WSYRK);
    if (st == LRVB_OK && gfree_out) st = d2h(c, gfree_out, gfr.p, (size_t)(N * KM));        // rank-local: this rank's rows
    // Everything that is a SUM OVER OBSERVATIONS goes into one device buffer, [S64 (4096) | val2 (2) | count of
    // non-positive-definite rows (1) | pad (1) | packed R (ldk x lda)], and is handed to the sum-over-ranks hook ONCE,
    // before anything is copied out.  Every rank takes the same path up to that reduction whatever its own rows look
    // like (a rank that stopped early would leave the others waiting in the collective); the indefinite-row count is
    // judged after it, so all ranks fail together.
    const i64 TAIL = 4100;
    const bool want_R = (R_out != nullptr) || (mx_flags & 1);
    if (st == LRVB_OK) st = buf_reserve(c, Rd, (size_t)(TAIL + (want_R ? ldk * lda : 0)));
    double* tail = Rd.p;
    double* Rpk = Rd.p + TAIL;
    // S64 = U^T diag(w) U
    if (st == LRVB_OK) st = reserve_obs_vec(c, c->zbuf);
    if (st == LRVB_OK && hipMemcpyAsync(c->zbuf.p, c->w.p, (size_t)N * sizeof(double), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) { lrvb_set_error("copy failed"); st = LRVB_ERR_HIP; }
    if (st == LRVB_OK) st = buf_reserve(c, c->Tdense, (size_t)WS_TILE * WS_TILE);
    if (st == LRVB_OK) st = launch_gram_small_on(c, U.p, N, 64, c->zbuf.p, c->Tdense.p);
    if (st == LRVB_OK) st = launch_tiles_to_dense(c, c->Tdense.p, 64, tail, 64, 0, 0, false);
    if (st == LRVB_OK) { EW(mixture_tail_kernel, (i64)4, (const double*)c->scal.p, (const int*)bad, tail + 4096); }
    if (st == LRVB_OK && want_R) {
        const bool onchip = (V == 31 && K == 32);        // 528 x N x 528: the x~ (x) x~ operand is generated inside the GEMM
        if (st == LRVB_OK && !onchip) st = buf_reserve(c, Xk, (size_t)((N + 16) * ldk));
        if (st == LRVB_OK && !onchip) st = launch_kron_rows(c, Xk.p, ldk);
        if (st == LRVB_OK) { EW(fill_kernel, N, 1.0, c->zbuf.p); }
        if (st == LRVB_OK && ((!onchip && hipMemsetAsync(Xk.p + N * ldk, 0, (size_t)(16 * ldk) * sizeof(double), c->stream) != hipSuccess) ||
                              hipMemsetAsync(Amat.p + N * lda, 0, (size_t)(16 * lda) * sizeof(double), c->stream) != hipSuccess)) {
            lrvb_set_error("memset failed"); st = LRVB_ERR_HIP;
        }
        if (st == LRVB_OK) st = onchip ? launch_atb_kron32(c, c->X.p, Amat.p, N, c->zbuf.p, Rpk)
                                       : launch_atb(c, Xk.p, ldk, Amat.p, lda, N, c->zbuf.p, Rpk, true);
    }
    if (st == LRVB_OK) st = obs_reduce(c, Rd.p, TAIL + (want_R ? ldk * lda : 0));
    double htail[4] = {0.0, 0.0, 0.0, 0.0};
    if (st == LRVB_OK) st = d2h(c, htail, tail + 4096, 4);
    if (st == LRVB_OK) { val2_out[0] = htail[0]; val2_out[1] = htail[1]; }
    if (st == LRVB_OK) st = d2h(c, S64_out, tail, 64 * 64);
    if (st == LRVB_OK && want_R) {
        if (htail[2] != 0.0) { lrvb_set_error("a local (simplex) Hessian block is not positive definite: the Schur complement is undefined at this point"); st = LRVB_ERR_NOT_POSDEF; }
        // Amat is dead now: reuse it for the expanded (V+1)^2 x K^2 result
        if (st == LRVB_OK) st = buf_reserve(c, Amat, (size_t)(QQ * KK));
        if (st == LRVB_OK) st = launch_mixture_expand(c, Rpk, lda, V + 1, K, Amat.p);
        if (st == LRVB_OK && !(mx_flags & 1)) st = d2h(c, R_out, Amat.p, (size_t)(QQ * KK));
        if (st == LRVB_OK) { c->mx_res_K = K; c->mx_res_q = V + 1; }
    }
    return st;
}

extern "C" int lrvb_mixture_rows(lrvb_ctx* c, int32_t K, const double* theta_z, const double* Lam,
                                 double* val2_out, double* gfree_out, double* S64_out, double* R_out) {
    return mixture_rows_impl(c, K, theta_z, Lam, val2_out, gfree_out, S64_out, R_out, 0);
}
extern "C" int lrvb_mixture_stats(lrvb_ctx* c, int32_t K, const double* theta_z, const double* Lam, int32_t want_schur,
                                  double* val2_out, double* S64_out) {
    return mixture_rows_impl(c, K, theta_z, Lam, val2_out, nullptr, S64_out, nullptr, want_schur ? 1 : 0);
}

// ---- Schur complement of the mixture's global block, assembled on the device -------------------------

// Shared tail of the two Schur entry points: Jd (n x n) and Hd (n x n) are on the device, sc / dg (nullable) too.
static int mixture_schur_core(lrvb_ctx* c, int32_t K, int32_t q, const double* sc, const double* dg, double* H_out) {
    const i64 n = (i64)K * q, nn = n * n;
    DevBuf &Rfull = c->mx_A, &Rm = c->mx_Xk, &Jd = c->mx_U, &T = c->mx_R, &Hd = c->mx_g;
    LRVB_TRY(buf_reserve(c, Rm, (size_t)nn));
    EW(mixture_permute_kernel, nn, (int)q, (int)K, Rfull.p, Rm.p);
    LRVB_TRY(buf_reserve(c, T, (size_t)nn));
    LRVB_TRY(gemm_tn(c, n, n, n, Rm.p, Jd.p, T.p));              // T = Rm^T J (Rm is symmetric up to rounding)
    LRVB_TRY(gemm_tn(c, n, n, n, Jd.p, T.p, Rm.p));              // S = J^T T, into the dead Rm
    EW(mixture_schur_finish_kernel, nn, n, (const double*)Hd.p, sc, dg, (const double*)Rm.p, c->Hfree.p);
    return H_out ? d2h(c, H_out, c->Hfree.p, (size_t)nn) : LRVB_OK;
}

extern "C" int lrvb_mixture_schur(lrvb_ctx* c, int32_t K, int32_t q, const double* R, const double* Jlam, const double* Hgg,
                                  const double* scale, const double* diag_add, double* H_out) {
    LRVB_TRY(ctx_bind(c));
    if (!Jlam || !Hgg || !H_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    if (K < 1 || q < 1 || (i64)K * q > 8192) LRVB_FAIL(LRVB_ERR_INVALID, "K, q out of range");
    const i64 n = (i64)K * q, nn = n * n;
    DevBuf &Rfull = c->mx_A, &Jd = c->mx_U, &Hd = c->mx_g;
    c->x2_ready = false;
    if (R) {
        LRVB_TRY(buf_reserve(c, Rfull, (size_t)nn));
        LRVB_TRY(h2d(c, Rfull.p, R, (size_t)nn));
        c->mx_res_K = K; c->mx_res_q = q;
    } else if (c->mx_res_K != K || c->mx_res_q != q) {
        LRVB_FAIL(LRVB_ERR_STATE, "R is NULL and no lrvb_mixture_rows result of this shape is resident");
    }
    LRVB_TRY(buf_reserve(c, Jd, (size_t)nn));
    LRVB_TRY(h2d(c, Jd.p, Jlam, (size_t)nn));
    // Hgg, scale and diag go through the Hessian scratch
    LRVB_TRY(buf_reserve(c, Hd, (size_t)(nn)));
    LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)(nn + 2 * n)));
    LRVB_TRY(h2d(c, Hd.p, Hgg, (size_t)nn));
    double* sc = nullptr; double* dg = nullptr;
    if (scale) { sc = c->Hfree.p + nn; LRVB_TRY(h2d(c, sc, scale, (size_t)n)); }
    if (diag_add) { dg = c->Hfree.p + nn + n; LRVB_TRY(h2d(c, dg, diag_add, (size_t)n)); }
    return mixture_schur_core(c, K, q, sc, dg, H_out);
}


// The same Schur complement with BOTH n x n inputs generated on the device from their O(n) description (six n-vectors
// and two (K + 1)-vectors instead of two 8 MB matrices at n = 1024), the operand R taken from the lrvb_mixture_rows /
// lrvb_mixture_stats call before it, and the result left on the device (H_out nullable; lrvb_chol_factor_last factors it).
// Parameter order: the K weights first, then the (V, K) array row-major.  vecs = [dl_diag | h_diag | scale | diag_add]
// (4 n), consts = [dl_const | h_const] (2 (K + 1)).
extern "C" int lrvb_mixture_schur_dirichlet(lrvb_ctx* c, int32_t K, int32_t q, const double* vecs, const double* consts, double* H_out) {
    LRVB_TRY(ctx_bind(c));
    if (!vecs || !consts) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    if (K < 1 || q < 1 || (i64)K * q > 8192) LRVB_FAIL(LRVB_ERR_INVALID, "K, q out of range");
    if (c->mx_res_K != K || c->mx_res_q != q) LRVB_FAIL(LRVB_ERR_STATE, "no lrvb_mixture_rows / lrvb_mixture_stats result of this shape is resident");
    const i64 n = (i64)K * q, nn = n * n;
    if (c->D != n) LRVB_FAIL(LRVB_ERR_SIZE, "the context's layout has %lld free parameters, the Dirichlet blocks %lld", (long long)c->D, (long long)n);
    c->x2_ready = false;
    const i64 nv = 4 * n + 2 * (K + 1), nb = K + 1;
    LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)(nn + nv)));
    double* v = c->Hfree.p + nn;
    DevBuf &Rfull = c->mx_A, &Rm = c->mx_Xk, &Sums = c->mx_R;
    LRVB_TRY(buf_reserve(c, Rm, (size_t)nn));
    LRVB_TRY(buf_reserve(c, Sums, (size_t)(2 * n * nb + nb * nb)));
    {   // one upload
        std::vector<double> pack((size_t)nv);
        memcpy(pack.data(), vecs, (size_t)(4 * n) * sizeof(double));
        memcpy(pack.data() + 4 * n, consts, (size_t)(2 * (K + 1)) * sizeof(double));
        LRVB_TRY(h2d(c, v, pack.data(), (size_t)nv));
    }
    const double* dl_diag = v; const double* h_diag = v + n; const double* sc = v + 2 * n; const double* dg = v + 3 * n;
    const double* dl_const = v + 4 * n; const double* h_const = dl_const + (K + 1);
    // J^T R J for the "diagonal + constant per Dirichlet" Jacobian in O(n^2) (k_models.hip): block row / column sums of R and the
    // (K + 1)^2 sums of those, then one element-wise kernel that also adds the Dirichlet Hessian blocks -- neither matrix is formed
    double* RS = Sums.p; double* SR = RS + n * nb; double* SRS = SR + n * nb;
    EW(mixture_permute_kernel, nn, (int)q, (int)K, Rfull.p, Rm.p);
    hipLaunchKernelGGL(dirichlet_rowsums_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, c->stream, n, (int)K, (const double*)Rm.p, RS);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(dirichlet_colsums_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)nb), dim3(256), 0, c->stream, n, (int)K, (const double*)Rm.p, SR);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(dirichlet_blocksums_kernel, dim3((unsigned)nb), dim3(64), 0, c->stream, n, (int)K, (const double*)RS, SRS);
    HIP_TRY(hipGetLastError());
    EW(dirichlet_schur_finish_kernel, nn, n, (int)K, (const double*)Rm.p, (const double*)RS, (const double*)SR, (const double*)SRS,
       dl_diag, dl_const, h_diag, h_const, sc, dg, c->Hfree.p);
    return H_out ? d2h(c, H_out, c->Hfree.p, (size_t)nn) : LRVB_OK;
}

// Gram matrix of per-observation gradients g_n[k] = 1/2 z_n^T M_k z_n + c_k, in FREE coordinates:
//   G^T G = J^T ( 1/4 M~^T K4 M~ + 1/2 (t c^T + c t^T) + N c c^T ) J,   t = M~^T s,
// K4 = sum_n (z_n (x) z_n)(z_n (x) z_n)^T from the Kronecker-row MFMA kernel, s = vec(sum_n z_n z_n^T),
// C (PA x PB) = A^T B for row-major A (K x PA), B (K x PB): the two-operand LDS-DMA MFMA kernel with unit
// contraction weights (~66 TFLOP/s at 4096^3) when the operands are even-width and 16-byte aligned,
// the generic 64 x 64 tile GEMM otherwise.
static int gemm_tn(lrvb_ctx* c, i64 K, i64 PA, i64 PB, const double* A, const double* B, double* C) {
    const bool big = K >= 512 && PA >= 128 && PB >= 128;
    if (big && ((PA % 2) || (PB % 2) || (((uintptr_t)A) & 15) || (((uintptr_t)B) & 15))) {
        // odd widths (the 995 global parameters of config 4): the MFMA kernel wants even, 16-byte aligned rows.  Copy the
        // operands into zero-padded even-width scratch (three 8 MB copies at n = 995: ~15 us) instead of running the
        // generic 64 x 64 tile GEMM (82 us per product there).
        const i64 PAe = PA + (PA & 1), PBe = PB + (PB & 1);
        LRVB_TRY(buf_reserve(c, c->gpad, (size_t)(K * PAe + K * PBe + PAe * PBe)));
        double* Ap = c->gpad.p; double* Bp = Ap + K * PAe; double* Cp = Bp + K * PBe;
        HIP_TRY(hipMemsetAsync(Ap, 0, (size_t)(K * PAe + K * PBe) * sizeof(double), c->stream));
        HIP_TRY(hipMemcpy2DAsync(Ap, (size_t)PAe * 8, A, (size_t)PA * 8, (size_t)PA * 8, (size_t)K, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpy2DAsync(Bp, (size_t)PBe * 8, B, (size_t)PB * 8, (size_t)PB * 8, (size_t)K, hipMemcpyDeviceToDevice, c->stream));
        LRVB_TRY(gemm_tn(c, K, PAe, PBe, Ap, Bp, Cp));
        HIP_TRY(hipMemcpy2DAsync(C, (size_t)PB * 8, Cp, (size_t)PBe * 8, (size_t)PB * 8, (size_t)PA, hipMemcpyDeviceToDevice, c->stream));
        return LRVB_OK;
    }
    const bool fast = big && !(PA % 2) && !(PB % 2) && !(((uintptr_t)A) & 15) && !(((uintptr_t)B) & 15);
    if (!fast) {
        if (PA <= 512 && PB <= 512 && K <= 4096 && PA * PB >= 4096) return launch_gemm_tn_small(c, K, PA, PB, A, B, C);   // a few hundred wide: 16 x 16 tiles fill the chip
        return launch_gemm(c, true, false, PA, PB, K, 1.0, A, PA, B, PB, 0.0, C, PB);
    }
    if (c->ones_n != K) {                      // the kernel reads up to 32 weights past K: they must be zero
        LRVB_TRY(buf_reserve(c, c->ones, (size_t)(K + 64)));
        EW(fill_kernel, K, 1.0, c->ones.p);
        HIP_TRY(hipMemsetAsync(c->ones.p + K, 0, 64 * sizeof(double), c->stream));
        c->ones_n = K;
    }
    return launch_atb(c, A, PA, B, PB, K, c->ones.p, C);
}


// The per-coordinate matrices M_k of the Wishart + MVN model's per-observation term (LRVB/NormalParams.py:6-23,
// WishartParams.py:6-35;  l_n = 1/2 z^T Q z + c with z = [y; 1], Q = nu [[V, -V m], [-m^T V, m^T V m]]), written on the device
// from (nu, m, V m, V): V q^2 doubles (134 MB at d = 63) that the host used to build and send over PCIe in every call.

static int quadform_gram_impl(lrvb_ctx* c, const double* M, const WishartGen* gen, const double* cvec, int64_t K,
                              const double* free_in, double* GtG_out, int64_t ld);
extern "C" int lrvb_quadform_gram(lrvb_ctx* c, const double* M, const double* cvec, int64_t K,
                                  const double* free_in, double* GtG_out, int64_t ld) {
    LRVB_TRY(ctx_bind(c));
    if (!M || !cvec || !free_in || !GtG_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    return quadform_gram_impl(c, M, nullptr, cvec, K, free_in, GtG_out, ld);
}
extern "C" int lrvb_wishart_gram(lrvb_ctx* c, int64_t d, const int64_t* offsets, double nu, const double* m, const double* v,
                                 const double* cvec, const double* free_in, double* GtG_out, int64_t ld) {
    LRVB_TRY(ctx_bind(c));
    if (!offsets || !m || !v || !cvec || !free_in) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    const i64 mm = d * (d + 1) / 2;
    if (d < 1 || d + 1 != c->P) LRVB_FAIL(LRVB_ERR_SIZE, "the data rows are [y; 1]: expected n_cols = d + 1 = %lld, got %lld", (long long)(d + 1), (long long)c->P);
    const i64 ms = offsets[0], ls = offsets[1], inu = offsets[2], vs = offsets[3];
    if (ms < 0 || ms + d > c->V || ls < 0 || ls + mm > c->V || inu < 0 || inu >= c->V || vs < 0 || vs + mm > c->V)
        LRVB_FAIL(LRVB_ERR_INVALID, "parameter offsets outside the %lld vector coordinates", (long long)c->V);
    // (m, V m, V) -> the device, in stream order (small uploads: no synchronisation)
    std::vector<double> pack((size_t)(2 * d + d * d));
    double mvm = 0.0;
    for (i64 i = 0; i < d; ++i) {
        double t = 0.0;
        for (i64 j = 0; j < d; ++j) t += v[i * d + j] * m[j];
        pack[(size_t)i] = m[i]; pack[(size_t)(d + i)] = t; mvm += m[i] * t;
    }
    memcpy(pack.data() + 2 * d, v, (size_t)(d * d) * sizeof(double));
    LRVB_TRY(buf_reserve(c, c->vtmp, pack.size() > (size_t)(c->V > c->D ? c->V : c->D) ? pack.size() : (size_t)(c->V > c->D ? c->V : c->D)));
    LRVB_TRY(h2d(c, c->vtmp.p, pack.data(), pack.size()));
    WishartGen gen{ d, ms, ls, inu, vs, nu, mvm, c->vtmp.p, c->vtmp.p + d, c->vtmp.p + 2 * d };
    return quadform_gram_impl(c, nullptr, &gen, cvec, c->V, free_in, GtG_out, ld);
}
static int quadform_gram_impl(lrvb_ctx* c, const double* M, const WishartGen* gen, const double* cvec, int64_t K,
                              const double* free_in, double* GtG_out, int64_t ld) {
    if (c->loss == LRVB_LOSS_NONE || !c->have_X) LRVB_FAIL(LRVB_ERR_STATE, "no data matrix: call lrvb_set_data(LRVB_SLOT_X) first");
    if (K != c->V) LRVB_FAIL(LRVB_ERR_SIZE, "expected one matrix per vector coordinate (%lld), got %lld", (long long)c->V, (long long)K);
    if (c->P > 64) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "Kronecker Gram kernel supports n_cols <= 64");
    if (ld < c->D) LRVB_FAIL(LRVB_ERR_SIZE, "leading dimension too small");
    const int q = (int)c->P;
    const i64 V = c->V, D = c->D, Pv = (i64)q * (q + 1) / 2;      // packed lower triangle of z z^T
    const int nbk = (int)((Pv + WS_TILE - 1) / WS_TILE);
    const i64 Pv_t = (i64)nbk * WS_TILE;                 // tile-padded virtual dimension (>= Pv)
    // unweighted sums: c_n = 1 (zero padded)
    LRVB_TRY(reserve_obs_vec(c, c->zbuf));
    EW(fill_kernel, c->N, 1.0, c->zbuf.p);
    // s = vec(Z^T Z) through the narrow Gram kernel
    LRVB_TRY(buf_reserve(c, c->stats, 1 + (size_t)c->P + (size_t)WS_TILE * WS_TILE));
    double* tile0 = c->stats.p + 1 + c->P;
    LRVB_TRY(launch_wsyrk(c, c->zbuf.p, tile0));
    // the three sums over observations -- the K4 tiles, s (q x q) and the observation count -- share one buffer and go
    // to the sum-over-ranks hook ONCE: [K4 tiles | s | N]
    const size_t tiles_n = (size_t)nbk * (nbk + 1) / 2 * WS_TILE * WS_TILE;
    LRVB_TRY(buf_reserve(c, c->Tdense, tiles_n + (size_t)q * q + 2));
    double* sdense = c->Tdense.p + tiles_n;
    double* ncount = sdense + (size_t)q * q;
    LRVB_TRY(launch_tiles_to_dense(c, tile0, q, sdense, q, 0, 0, false));
    EW(fill_kernel, (i64)1, (double)c->N, ncount);
    // K4 (dense, Pv_t x Pv_t).  Every buffer of the call is reserved BEFORE the launch (an allocation may synchronise), and
    // the three host operands -- M is V q^2 doubles: 134 MB in configuration 5 -- are uploaded on the side stream WHILE
    // the Kronecker kernel runs (they used to wait for it on the context's stream and then cost 12 ms of a 287 ms step).
    LRVB_TRY(buf_reserve(c, c->vtmp2, (size_t)(Pv_t > V ? Pv_t : V)));
    LRVB_TRY(buf_reserve(c, c->Heta, (size_t)Pv_t * (size_t)Pv_t));
    if (M) LRVB_TRY(buf_reserve(c, c->work1, (size_t)V * (size_t)q * (size_t)q));
    LRVB_TRY(buf_reserve(c, c->Jdense, (size_t)Pv_t * (size_t)V > (size_t)V * (size_t)D ? (size_t)Pv_t * (size_t)V : (size_t)V * (size_t)D));
    LRVB_TRY(buf_reserve(c, c->vtmp3, (size_t)(V > D ? V : D)));
    LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)D * (size_t)D));
    DevBuf &Mt = c->qg_Mt, &T1 = c->qg_T1, &Av = c->qg_Av;       // kept in the context between calls
    if (M) LRVB_TRY(buf_reserve(c, Mt, (size_t)Pv_t * (size_t)V));
    LRVB_TRY(buf_reserve(c, T1, (size_t)(Pv_t > D ? Pv_t : D) * (size_t)V));      // K4 M~ (Pv_t x V), later J^T Av (D x V)
    LRVB_TRY(buf_reserve(c, Av, (size_t)V * (size_t)V));
    LRVB_TRY(launch_wsyrk_kron(c, c->zbuf.p, c->Tdense.p));
    if (M) LRVB_TRY(h2d_beside(c, c->work1.p, M, (size_t)V * (size_t)q * (size_t)q));       // 134 MB at configuration 5's size: beside the Kronecker kernel
    LRVB_TRY(h2d_beside(c, c->g_eta.p, cvec, (size_t)V));
    LRVB_TRY(h2d_beside(c, c->theta.p, free_in, (size_t)D));
    LRVB_TRY(obs_reduce(c, c->Tdense.p, (i64)(tiles_n + (size_t)q * q + 1)));
    HIP_TRY(hipMemsetAsync(c->vtmp2.p, 0, (size_t)Pv_t * sizeof(double), c->stream));
    hipLaunchKernelGGL(svec_kernel, dim3(nb256(Pv)), dim3(256), 0, c->stream, sdense, q, c->vtmp2.p);
    HIP_TRY(hipGetLastError());
    LRVB_TRY(launch_tiles_to_dense(c, c->Tdense.p, Pv_t, c->Heta.p, Pv_t, 0, 0, false));
    // T1 = K4 M~ ;  Av = M~^T T1 ;  t = M~^T s
    if (M) {
        // general matrices: M~ (Pv_t x V) written out, two products on the matrix cores
        HIP_TRY(hipMemsetAsync(Mt.p, 0, (size_t)Pv_t * (size_t)V * sizeof(double), c->stream));
        {
            dim3 grid(nb256(V), (unsigned)Pv);
            hipLaunchKernelGGL(mtilde_kernel, grid, dim3(256), 0, c->stream, c->work1.p, V, q, Mt.p);
            HIP_TRY(hipGetLastError());
        }
        LRVB_TRY(gemm_tn(c, Pv_t, Pv_t, V, c->Heta.p, Mt.p, T1.p));        // K4 is symmetric: K4 M~ = K4^T M~
        LRVB_TRY(gemm_tn(c, Pv_t, V, V, Mt.p, T1.p, Av.p));
        LRVB_TRY(launch_gemv(c, true, Pv_t, V, 1.0, Mt.p, V, c->vtmp2.p, 0.0, c->vtmp3.p));
    } else {
        // the Wishart + MVN model: M~ is 0.2 % dense and is never written -- every column is a gather of a few rows (k_models.hip)
        // (the one dense column, the coordinate nu, is a matrix-vector product with its Pv coefficients; they and the product
        // vector live in c->Jdense, which this route does not use otherwise)
        double* cnu = c->Jdense.p; double* tmpv = cnu + Pv_t;
        hipLaunchKernelGGL(wishart_nu_coef_kernel, dim3(nb256(Pv)), dim3(256), 0, c->stream, *gen, Pv, cnu);
        HIP_TRY(hipGetLastError());
        dim3 gr(nb256(V), (unsigned)Pv);
        hipLaunchKernelGGL(wishart_sparse_right_kernel, gr, dim3(256), 0, c->stream, *gen, Pv, V, (const double*)c->Heta.p, Pv_t, T1.p, V);
        HIP_TRY(hipGetLastError());
        LRVB_TRY(launch_gemv(c, false, Pv, Pv, 1.0, c->Heta.p, Pv_t, cnu, 0.0, tmpv));                  // K4 c_nu
        hipLaunchKernelGGL(scatter_column_kernel, dim3(nb256(Pv)), dim3(256), 0, c->stream, Pv, (const double*)tmpv, T1.p, V, gen->inu);
        HIP_TRY(hipGetLastError());
        dim3 gl(nb256(V), (unsigned)V);
        hipLaunchKernelGGL(wishart_sparse_left_kernel, gl, dim3(256), 0, c->stream, *gen, Pv, V, (const double*)T1.p, V, Av.p, V);
        HIP_TRY(hipGetLastError());
        LRVB_TRY(launch_gemv(c, true, Pv, V, 1.0, T1.p, V, cnu, 0.0, Av.p + gen->inu * V));             // row nu of M~^T T1
        dim3 gt(1, (unsigned)V);
        hipLaunchKernelGGL(wishart_sparse_left_kernel, gt, dim3(256), 0, c->stream, *gen, Pv, (i64)1, (const double*)c->vtmp2.p, (i64)1, c->vtmp3.p, (i64)1);
        HIP_TRY(hipGetLastError());
        LRVB_TRY(launch_gemv(c, true, Pv, 1, 1.0, c->vtmp2.p, 1, cnu, 0.0, c->vtmp3.p + gen->inu));     // t[nu] = c_nu . s
    }
    {
        dim3 grid(nb256(V), (unsigned)V);
        hipLaunchKernelGGL(rank_terms_kernel, grid, dim3(256), 0, c->stream, V, (const double*)ncount, c->vtmp3.p, c->g_eta.p, Av.p);
        HIP_TRY(hipGetLastError());
    }
    // free coordinates: J^T Av J
    if (c->jt_rows > 0) {                                                             // two structured products (k_pack.hip): no dense Jacobian, no 2 V^2 D products
        LRVB_TRY(launch_jt_apply(c, c->theta.p, Av.p, V, V, T1.p, V, false));          // J^T Av   (D x V)
        LRVB_TRY(launch_jt_apply(c, c->theta.p, T1.p, V, D, c->Hfree.p, D, true));     // J^T (J^T Av)^T
    } else {
        LRVB_TRY(launch_dense_jac(c, c->theta.p, c->Jdense.p));
        LRVB_TRY(gemm_tn(c, V, V, D, Av.p, c->Jdense.p, T1.p));         // Av is symmetric
        LRVB_TRY(gemm_tn(c, V, D, D, c->Jdense.p, T1.p, c->Hfree.p));
    }
    if (GtG_out) {
        HIP_TRY(hipMemcpy2DAsync(GtG_out, (size_t)ld * 8, c->Hfree.p, (size_t)D * 8, (size_t)D * 8, (size_t)D, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return LRVB_OK;
}

// Conjugate gradients on a dense symmetric matrix held on the device (objectives whose Hessian is
// assembled from sufficient statistics: the HVP is a D x D matrix-vector product).  H == NULL reuses
// the matrix of the previous call.  Same stopping rule as lrvb_cg_solve.
extern "C" int lrvb_cg_solve_matrix(lrvb_ctx* c, const double* H, const double* b, const double* x0,
                                    const double* Minv, double tol, int64_t maxiter, int64_t D,
                                    double* x_out, int* info_out, int64_t* iters_out) {
    LRVB_TRY(ctx_bind(c));
    if (!b || !x_out || D <= 0) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    if (H) {
        LRVB_TRY(buf_reserve(c, c->cgH, (size_t)D * (size_t)D));
        LRVB_TRY(h2d(c, c->cgH.p, H, (size_t)D * (size_t)D));
        c->cgH_n = D;
    } else if (c->cgH_n != D) {
        LRVB_FAIL(LRVB_ERR_STATE, "no %lld x %lld matrix resident: pass H once", (long long)D, (long long)D);
    }
    if (maxiter <= 0) maxiter = 10 * D;
    DevBuf* vecs[] = { &c->rhs, &c->cgx, &c->cgr, &c->cgp, &c->cgq, &c->cgz };
    for (DevBuf* v : vecs) LRVB_TRY(buf_reserve(c, *v, (size_t)D));
    if (Minv) { LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)D * (size_t)D)); LRVB_TRY(h2d(c, c->Hfree.p, Minv, (size_t)D * (size_t)D)); }
    LRVB_TRY(h2d(c, c->rhs.p, b, (size_t)D));
    double* s = c->scal.p;
    double hs[4];
    LRVB_TRY(launch_dot(c, c->rhs.p, c->rhs.p, D, s + 0));
    if (x0) {
        LRVB_TRY(h2d(c, c->cgx.p, x0, (size_t)D));
        LRVB_TRY(launch_gemv(c, false, D, D, 1.0, c->cgH.p, D, c->cgx.p, 0.0, c->cgq.p));
        LRVB_TRY(launch_axpby(c, D, 1.0, c->rhs.p, 0.0, c->cgr.p));
        LRVB_TRY(launch_axpby(c, D, -1.0, c->cgq.p, 1.0, c->cgr.p));
    } else {
        HIP_TRY(hipMemsetAsync(c->cgx.p, 0, (size_t)D * sizeof(double), c->stream));
        LRVB_TRY(launch_axpby(c, D, 1.0, c->rhs.p, 0.0, c->cgr.p));
    }
    LRVB_TRY(d2h(c, hs, s, 1));
    const double bnorm = sqrt(hs[0]);
    const double atol = tol * bnorm;
    int info = 0; i64 it = 0;
    double rho_prev = 0.0;
    if (bnorm == 0.0) {
        HIP_TRY(hipMemsetAsync(c->cgx.p, 0, (size_t)D * sizeof(double), c->stream));
    } else {
        info = (int)maxiter;
        for (it = 0; it < maxiter; ++it) {
            if (Minv) LRVB_TRY(launch_gemv(c, false, D, D, 1.0, c->Hfree.p, D, c->cgr.p, 0.0, c->cgz.p));
            const double* z = Minv ? c->cgz.p : c->cgr.p;
            LRVB_TRY(launch_dot(c, c->cgr.p, c->cgr.p, D, s + 1));
            LRVB_TRY(launch_dot(c, c->cgr.p, z, D, s + 2));
            LRVB_TRY(d2h(c, hs + 1, s + 1, 2));
            if (sqrt(hs[1]) < atol) { info = 0; break; }
            const double rho = hs[2];
            if (it > 0) LRVB_TRY(launch_axpby(c, D, 1.0, z, rho / rho_prev, c->cgp.p));
            else        LRVB_TRY(launch_axpby(c, D, 1.0, z, 0.0, c->cgp.p));
            LRVB_TRY(launch_gemv(c, false, D, D, 1.0, c->cgH.p, D, c->cgp.p, 0.0, c->cgq.p));
            LRVB_TRY(launch_dot(c, c->cgp.p, c->cgq.p, D, s + 3));
            LRVB_TRY(d2h(c, hs + 3, s + 3, 1));
            const double alpha = rho / hs[3];
            LRVB_TRY(launch_axpby(c, D, alpha, c->cgp.p, 1.0, c->cgx.p));
            LRVB_TRY(launch_axpby(c, D, -alpha, c->cgq.p, 1.0, c->cgr.p));
            rho_prev = rho;
        }
    }
    LRVB_TRY(d2h(c, x_out, c->cgx.p, (size_t)D));
    if (info_out) *info_out = info;
    if (iters_out) *iters_out = it;
    return LRVB_OK;
}

// ---- Cholesky / linear response ------------------------------------------------------------
static int chol_factor_dev_impl(lrvb_ctx* c, const double* H_dev, i64 D, i64 ld) {
    LRVB_TRY(buf_reserve(c, c->chol, (size_t)D * (size_t)D));
    HIP_TRY(hipMemcpy2DAsync(c->chol.p, (size_t)D * 8, H_dev, (size_t)ld * 8, (size_t)D * 8, (size_t)D, hipMemcpyDeviceToDevice, c->stream));
    int* info = reinterpret_cast<int*>(c->scal.p);
    c->chol_valid = false;
    LRVB_TRY(launch_potrf_lower(c, c->chol.p, D, D, info));
    int h = 0;
    HIP_TRY(hipMemcpyAsync(&h, info, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h != 0) LRVB_FAIL(LRVB_ERR_NOT_POSDEF, "%d-th leading minor of the array is not positive definite", h);
    c->chol_valid = true; c->chol_n = D;
    return LRVB_OK;
}
extern "C" int lrvb_chol_factor_dev(lrvb_ctx* c, const double* H_dev, int64_t D, int64_t ld) {
    LRVB_TRY(ctx_bind(c));
    if (!H_dev || D <= 0 || ld < D) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    return chol_factor_dev_impl(c, H_dev, D, ld);
}
extern "C" int lrvb_chol_factor(lrvb_ctx* c, const double* H, int64_t D) {
    LRVB_TRY(ctx_bind(c));
    if (!H || D <= 0) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)D * (size_t)D));
    LRVB_TRY(h2d(c, c->Hfree.p, H, (size_t)D * (size_t)D));
    return chol_factor_dev_impl(c, c->Hfree.p, D, D);
}
extern "C" int lrvb_chol_factor_last(lrvb_ctx* c) {
    LRVB_TRY(ctx_bind(c));
    if (!c->Hfree.p || c->Hfree.n < (size_t)c->D * (size_t)c->D) LRVB_FAIL(LRVB_ERR_STATE, "no Hessian has been built through the host API yet");
    return chol_factor_dev_impl(c, c->Hfree.p, c->D, c->D);
}
extern "C" int lrvb_chol_solve_dev(lrvb_ctx* c, double* B_dev, int64_t D, int64_t nrhs) {
    LRVB_TRY(ctx_bind(c));
    if (!c->chol_valid || c->chol_n != D) LRVB_FAIL(LRVB_ERR_STATE, "no Cholesky factor of size %lld: call lrvb_chol_factor first", (long long)D);
    if (!B_dev || nrhs <= 0) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    return launch_potrs_lower(c, c->chol.p, D, D, B_dev, nrhs, nrhs);
}
extern "C" int lrvb_chol_solve(lrvb_ctx* c, const double* B, int64_t D, int64_t nrhs, double* X_out) {
    LRVB_TRY(ctx_bind(c));
    if (!c->chol_valid || c->chol_n != D) LRVB_FAIL(LRVB_ERR_STATE, "no Cholesky factor of size %lld: call lrvb_chol_factor first", (long long)D);
    if (!B || !X_out || nrhs <= 0) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    LRVB_TRY(buf_reserve(c, c->rhs, (size_t)D * (size_t)nrhs));
    LRVB_TRY(h2d(c, c->rhs.p, B, (size_t)D * (size_t)nrhs));
    LRVB_TRY(launch_potrs_lower(c, c->chol.p, D, D, c->rhs.p, nrhs, nrhs));
    return d2h(c, X_out, c->rhs.p, (size_t)D * (size_t)nrhs);
}
static int gemm_tn(lrvb_ctx* c, i64 K, i64 PA, i64 PB, const double* A, const double* B, double* C);
static int lrvb_cov_dev_impl(lrvb_ctx* c, const double* M_dev, i64 Q, i64 D, double* cov_dev) {
    if (!c->chol_valid || c->chol_n != D) LRVB_FAIL(LRVB_ERR_STATE, "no Cholesky factor of size %lld: call lrvb_chol_factor first", (long long)D);
    LRVB_TRY(buf_reserve(c, c->rhs, (size_t)D * (size_t)Q));
    dim3 grid(nb256(D), (unsigned)Q);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, c->stream, Q, D, M_dev, c->rhs.p);   // rhs = M^T (D x Q)
    HIP_TRY(hipGetLastError());
    // M H^-1 M^T = Y^T Y with Y = L^-1 M^T: one forward substitution and a TN product -- half the
    // triangular work of solve-then-multiply, and the result is symmetric by construction
    LRVB_TRY(launch_trsm_lower_forward(c, c->chol.p, D, D, c->rhs.p, Q, Q));
    return gemm_tn(c, D, Q, Q, c->rhs.p, c->rhs.p, cov_dev);
}
extern "C" int lrvb_lrvb_cov_dev(lrvb_ctx* c, const double* M_dev, int64_t Q, int64_t D, double* cov_dev) {
    LRVB_TRY(ctx_bind(c));
    if (!M_dev || !cov_dev || Q <= 0) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    return lrvb_cov_dev_impl(c, M_dev, Q, D, cov_dev);
}
extern "C" int lrvb_lrvb_cov(lrvb_ctx* c, const double* M, int64_t Q, int64_t D, double* cov_out) {
    LRVB_TRY(ctx_bind(c));
    if (!M || !cov_out || Q <= 0) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)Q * (size_t)D));
    LRVB_TRY(buf_reserve(c, c->Heta, (size_t)Q * (size_t)Q));
    LRVB_TRY(h2d(c, c->work1.p, M, (size_t)Q * (size_t)D));
    LRVB_TRY(lrvb_cov_dev_impl(c, c->work1.p, Q, D, c->Heta.p));
    return d2h(c, cov_out, c->Heta.p, (size_t)Q * (size_t)Q);
}

// ---- weight sensitivity of moments, streamed over the observations ------------
static int obs_influence_impl(lrvb_ctx* c, const double* point, i64 n_in, bool is_free, const double* M, i64 Q,
                              i64 n0, i64 n1, double* out) {
    LRVB_TRY(ctx_bind(c));
    if (!point || !M || !out || Q <= 0) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    const i64 width = is_free ? c->D : c->V;
    LRVB_TRY(check_len(n_in, width, is_free ? "free vector" : "vector"));
    if (c->loss == LRVB_LOSS_NONE || c->data_only) LRVB_FAIL(LRVB_ERR_STATE, "model has no declared data term");
    if (n0 < 0 || n1 > c->N || n0 > n1) LRVB_FAIL(LRVB_ERR_INVALID, "row range [%lld, %lld) outside [0, %lld)", (long long)n0, (long long)n1, (long long)c->N);
    if (!c->chol_valid || c->chol_n != width) LRVB_FAIL(LRVB_ERR_STATE, "no Cholesky factor of size %lld: call lrvb_chol_factor first", (long long)width);
    LRVB_TRY(data_ready(c));
    LRVB_TRY(h2d(c, c->theta.p, point, (size_t)width));
    LRVB_TRY(set_point(c, c->theta.p, is_free));
    LRVB_TRY(eval_grad_eta(c, c->stats.p, false, false));          // leaves l'_n in c->lp (rank-local rows)
    // W = H^-1 M^T
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)Q * (size_t)width));
    LRVB_TRY(h2d(c, c->work1.p, M, (size_t)Q * (size_t)width));
    LRVB_TRY(buf_reserve(c, c->rhs, (size_t)width * (size_t)Q));
    {
        dim3 grid(nb256(width), (unsigned)Q);
        hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, c->stream, Q, width, c->work1.p, c->rhs.p);
        HIP_TRY(hipGetLastError());
    }
    LRVB_TRY(launch_potrs_lower(c, c->chol.p, width, width, c->rhs.p, Q, Q));
    // Z = J_glm W  (P x Q)
    LRVB_TRY(buf_reserve(c, c->Heta, (size_t)c->P * (size_t)Q));
    if (!is_free || c->all_box) {
        EW(scale_slice_rows_kernel, c->P * Q, Q, is_free ? c->j1.p + c->glm_off : nullptr, c->rhs.p + c->glm_off * Q, c->Heta.p);
    } else {
        LRVB_TRY(ensure_dense_J(c, c->theta.p));
        LRVB_TRY(launch_gemm(c, false, false, c->P, Q, c->D, 1.0, c->Jdense.p + c->glm_off * c->D, c->D, c->rhs.p, Q, 0.0, c->Heta.p, Q));
    }
    if (hvp_multi_supported(c, 1) && !c->force_generic_wsyrk && n1 > n0) {
        // streamed: 8-row chunks of X through LDS, 16 moments at a time on the matrix cores (k_hvp_multi.hip)
        const i64 rows = n1 - n0;
        LRVB_TRY(reserve_obs_vec(c, c->zbuf));                                   // -l'_n with zero padding past N
        HIP_TRY(hipMemcpyAsync(c->zbuf.p, c->lp.p, (size_t)c->N * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        LRVB_TRY(launch_axpby(c, c->N, 0.0, c->zbuf.p, -1.0, c->zbuf.p));
        LRVB_TRY(buf_reserve(c, c->vtmp3, (size_t)Q * (size_t)c->P));             // Z^T (Q x P)
        {
            dim3 grid(nb256(Q), (unsigned)c->P);
            hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, c->stream, c->P, Q, c->Heta.p, c->vtmp3.p);
            HIP_TRY(hipGetLastError());
        }
        LRVB_TRY(buf_reserve(c, c->work1, (size_t)rows * (size_t)Q));
        for (i64 q0 = 0; q0 < Q; q0 += 16) {
            const i64 qn = (Q - q0 < 16) ? Q - q0 : 16;
            LRVB_TRY(launch_rows_times_matrix(c, n0, n1, qn, c->vtmp3.p + q0 * c->P, c->P, c->zbuf.p, c->work1.p + q0, Q));
        }
        return d2h(c, out, c->work1.p, (size_t)rows * (size_t)Q);
    }
    const i64 chunk = 65536;
    for (i64 a = n0; a < n1; a += chunk) {
        const i64 b = (a + chunk < n1) ? a + chunk : n1;
        const i64 rows = b - a;
        LRVB_TRY(buf_reserve(c, c->work1, (size_t)rows * (size_t)Q));
        LRVB_TRY(launch_gemm(c, false, false, rows, Q, c->P, 1.0, c->X.p + a * c->P, c->P, c->Heta.p, Q, 0.0, c->work1.p, Q));
        EW(row_scale_rows_kernel, rows * Q, Q, c->lp.p + a, -1.0, c->work1.p);
        LRVB_TRY(d2h(c, out + (a - n0) * Q, c->work1.p, (size_t)rows * (size_t)Q));
    }
    return LRVB_OK;
}
extern "C" int lrvb_obs_influence(lrvb_ctx* c, const double* free_in, int64_t D, const double* M, int64_t Q,
                                  int64_t n0, int64_t n1, double* out) {
    return obs_influence_impl(c, free_in, D, true, M, Q, n0, n1, out);
}
extern "C" int lrvb_obs_influence_vec(lrvb_ctx* c, const double* vec_in, int64_t V, const double* M, int64_t Q,
                                      int64_t n0, int64_t n1, double* out) {
    return obs_influence_impl(c, vec_in, V, false, M, Q, n0, n1, out);
}

// ---- conjugate gradient ----------------------------------------------------------------------
extern "C" int lrvb_cg_solve(lrvb_ctx* c, const double* free_in, const double* b, const double* x0,
                             const double* Minv, double tol, int64_t maxiter, int64_t D,
                             double* x_out, int* info_out, int64_t* iters_out) {
    const bool reuse = same_point(c, free_in, D, true);          // many right-hand sides at one point: the state stays
    const bool prepared = reuse && c->hvp_pt_prepared;
    LRVB_TRY(ctx_bind(c));
    if (!free_in || !b || !x_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(check_len(D, c->D, "free vector"));
    LRVB_TRY(data_ready(c));
    if (maxiter <= 0) maxiter = 10 * D;
    DevBuf* vecs[] = { &c->rhs, &c->cgx, &c->cgr, &c->cgp, &c->cgq, &c->cgz };
    for (DevBuf* v : vecs) LRVB_TRY(buf_reserve(c, *v, (size_t)D));
    if (Minv) { LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)D * (size_t)D)); LRVB_TRY(h2d(c, c->Hfree.p, Minv, (size_t)D * (size_t)D)); }
    bool resident = false;                                 // the Hessian of this point is resident: products are D x D gemv's
    LRVB_TRY(hres_matches(c, free_in, D, &resident));
    LRVB_TRY(h2d(c, c->rhs.p, b, (size_t)D));
    // point state once: eta, J, g_eta, cached curvature
    if (!resident) {
        if (!reuse) {
            LRVB_TRY(h2d(c, c->theta.p, free_in, (size_t)D));
            LRVB_TRY(set_point(c, c->theta.p, true));
            LRVB_TRY(eval_grad_eta(c, c->stats.p, true));
            c->pt_products = 0;
        }
        if (!prepared) LRVB_TRY(prepare_general_hvp(c, c->theta.p));
    }
    auto product = [&](const double* vin, double* vout) -> int {
        if (!resident) {                                   // (the right-hand sides of one point, solved one by one: see maybe_build_resident)
            ++c->pt_products;
            LRVB_TRY(maybe_build_resident(c, free_in, D, &resident));
        }
        return resident ? launch_gemv(c, false, D, D, 1.0, c->Hres.p, D, vin, 0.0, vout) : hvp_apply(c, c->theta.p, true, vin, vout);
    };

    double* s = c->scal.p;                     // s[0] = ||b||^2, s[1] = ||r||^2, s[2] = r.z, s[3] = p.q
    double hs[4];
    LRVB_TRY(launch_dot(c, c->rhs.p, c->rhs.p, D, s + 0));
    if (x0) {
        LRVB_TRY(h2d(c, c->cgx.p, x0, (size_t)D));
        LRVB_TRY(product(c->cgx.p, c->cgq.p));
        LRVB_TRY(launch_axpby(c, D, 1.0, c->rhs.p, 0.0, c->cgr.p));
        LRVB_TRY(launch_axpby(c, D, -1.0, c->cgq.p, 1.0, c->cgr.p));
    } else {
        HIP_TRY(hipMemsetAsync(c->cgx.p, 0, (size_t)D * sizeof(double), c->stream));
        LRVB_TRY(launch_axpby(c, D, 1.0, c->rhs.p, 0.0, c->cgr.p));
    }
    LRVB_TRY(d2h(c, hs, s, 1));
    const double bnorm = sqrt(hs[0]);
    const double atol = tol * bnorm;
    int info = 0; i64 it = 0;
    double rho_prev = 0.0;
    if (bnorm == 0.0) {
        HIP_TRY(hipMemsetAsync(c->cgx.p, 0, (size_t)D * sizeof(double), c->stream));
    } else {
        info = (int)maxiter;
        for (it = 0; it < maxiter; ++it) {
            if (Minv) LRVB_TRY(launch_gemv(c, false, D, D, 1.0, c->Hfree.p, D, c->cgr.p, 0.0, c->cgz.p));
            const double* z = Minv ? c->cgz.p : c->cgr.p;
            LRVB_TRY(launch_dot(c, c->cgr.p, c->cgr.p, D, s + 1));
            LRVB_TRY(launch_dot(c, c->cgr.p, z, D, s + 2));
            LRVB_TRY(d2h(c, hs + 1, s + 1, 2));
            if (sqrt(hs[1]) < atol) { info = 0; break; }
            const double rho = hs[2];
            if (it > 0) LRVB_TRY(launch_axpby(c, D, 1.0, z, rho / rho_prev, c->cgp.p));
            else        LRVB_TRY(launch_axpby(c, D, 1.0, z, 0.0, c->cgp.p));
            LRVB_TRY(product(c->cgp.p, c->cgq.p));
            LRVB_TRY(launch_dot(c, c->cgp.p, c->cgq.p, D, s + 3));
            LRVB_TRY(d2h(c, hs + 3, s + 3, 1));
            const double alpha = rho / hs[3];
            LRVB_TRY(launch_axpby(c, D, alpha, c->cgp.p, 1.0, c->cgx.p));
            LRVB_TRY(launch_axpby(c, D, -alpha, c->cgq.p, 1.0, c->cgr.p));
            rho_prev = rho;
        }
    }
    LRVB_TRY(d2h(c, x_out, c->cgx.p, (size_t)D));
    if (info_out) *info_out = info;
    if (iters_out) *iters_out = it;
    if (!resident) remember_point(c, free_in, D, true, true);
    return LRVB_OK;
}

// ---- higher-order directional derivatives of the gradient (vector coordinates) -------------------------

// ---- per-observation loss values: the gradient of the objective with respect to the weights -----------------

extern "C" int lrvb_obs_loss(lrvb_ctx* c, const double* point, int64_t n_in, int is_free, int64_t n0, int64_t n1, double* out) {
    LRVB_TRY(ctx_bind(c));
    if (!point || !out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    const i64 width = is_free ? c->D : c->V;
    LRVB_TRY(check_len(n_in, width, is_free ? "free vector" : "vector"));
    if (c->loss == LRVB_LOSS_NONE || c->data_only) LRVB_FAIL(LRVB_ERR_STATE, "model has no declared data term");
    if (n0 < 0 || n1 > c->N || n0 > n1) LRVB_FAIL(LRVB_ERR_INVALID, "row range [%lld, %lld) outside [0, %lld)", (long long)n0, (long long)n1, (long long)c->N);
    LRVB_TRY(data_ready(c));
    const i64 rows = n1 - n0, P = c->P;
    if (rows == 0) return LRVB_OK;
    LRVB_TRY(h2d(c, c->theta.p, point, (size_t)width));
    LRVB_TRY(set_point(c, c->theta.p, is_free != 0));
    // z = X[n0:n1] beta: the one-vector case of the skinny row product
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)rows * 2));
    double* z = c->work1.p; double* lv = c->work1.p + rows;
    if (hvp_multi_supported(c, 1) && !c->force_generic_wsyrk) {
        LRVB_TRY(reserve_obs_vec(c, c->zbuf));
        EW(fill_kernel, c->N, 1.0, c->zbuf.p);
        LRVB_TRY(launch_rows_times_matrix(c, n0, n1, 1, c->eta.p + c->glm_off, P, c->zbuf.p, z, 1));
    } else {
        LRVB_TRY(launch_gemv(c, false, rows, P, 1.0, c->X.p + n0 * P, P, c->eta.p + c->glm_off, 0.0, z));
    }
    EW(obs_loss_kernel, rows, (int)c->loss, c->lik_info, (const double*)(c->y.p + n0), (const double*)z, lv);
    return d2h(c, out, lv, (size_t)rows);
}

// ---- non-conjugate logistic term by Gauss-Hermite quadrature (LRVB/Modeling.py:36-52) ------------------------------------

extern "C" int lrvb_gh_logistic(lrvb_ctx* c, int64_t n, const double* z_mean, const double* z_sd, const double* gh_x, const double* gh_w,
                                int32_t n_nodes, int32_t order, double* val, double* d1, double* d2) {
    LRVB_TRY(ctx_bind(c));
    if (n < 0 || !z_mean || !z_sd || !gh_x || !gh_w || !val) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    if (n_nodes < 1 || n_nodes > 128) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "1 to 128 quadrature nodes");
    if (order < 0 || order > 2 || (order >= 1 && !d1) || (order >= 2 && !d2)) LRVB_FAIL(LRVB_ERR_INVALID, "order 0..2 with its output arrays");
    if (n == 0) return LRVB_OK;
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)n * 8 + 256));
    double* zm = c->work1.p; double* zs = zm + n; double* v = zs + n; double* o1 = v + n; double* o2 = o1 + 2 * n; double* g = o2 + 3 * n;
    LRVB_TRY(h2d(c, zm, z_mean, (size_t)n));
    LRVB_TRY(h2d(c, zs, z_sd, (size_t)n));
    LRVB_TRY(h2d(c, g, gh_x, (size_t)n_nodes));
    LRVB_TRY(h2d(c, g + 128, gh_w, (size_t)n_nodes));
    EW(gh_logistic_kernel, n, (const double*)zm, (const double*)zs, (const double*)g, (const double*)(g + 128), (int)n_nodes, (int)order, v, o1, o2);
    LRVB_TRY(d2h(c, val, v, (size_t)n));
    if (order >= 1) LRVB_TRY(d2h(c, d1, o1, (size_t)n * 2));
    if (order >= 2) LRVB_TRY(d2h(c, d2, o2, (size_t)n * 3));
    return LRVB_OK;
}

// ---- logistic regression with a mean-field Gaussian variational posterior (the model that expectation is written for) -------
// C (P x P) = A^T diag(cvec) B over N rows: the LDS-DMA MFMA kernel when the operands allow, the generic tile GEMM on a
// row-scaled copy otherwise
static int weighted_tn(lrvb_ctx* c, const double* A, const double* B, i64 P, i64 N, const double* cvec_padded, double* C, DevBuf& scratch) {
    const bool fast = !(P % 2) && !(((uintptr_t)A) & 15) && !(((uintptr_t)B) & 15) && P >= 2;
    if (fast) return launch_atb(c, A, P, B, P, N, cvec_padded, C);
    LRVB_TRY(buf_reserve(c, scratch, (size_t)(N * P)));
    EW(rowscale_kernel, N * P, P, cvec_padded, B, scratch.p);
    return launch_gemm(c, true, false, P, P, N, 1.0, A, P, scratch.p, P, 0.0, C, P);
}

extern "C" int lrvb_logitnormal_terms(lrvb_ctx* c, const double* mean, const double* var, int64_t P_in, const double* gh_x,
                                      const double* gh_w, int32_t n_nodes, double* value_out, double* grad_out, double* H_blocks_out) {
    LRVB_TRY(ctx_bind(c));
    if (!mean || !var || !gh_x || !gh_w || !value_out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    if (n_nodes < 1 || n_nodes > 128) LRVB_FAIL(LRVB_ERR_UNSUPPORTED, "1 to 128 quadrature nodes");
    if (c->loss == LRVB_LOSS_NONE || c->data_only || !(c->have_X && c->have_y))
        LRVB_FAIL(LRVB_ERR_STATE, "the context needs a design matrix and responses: lrvb_set_data for LRVB_SLOT_X and LRVB_SLOT_Y");
    const i64 N = c->N, P = c->P;
    LRVB_TRY(check_len(P_in, P, "mean / var"));
    for (i64 j = 0; j < P; ++j) if (!(var[j] > 0.0)) LRVB_FAIL(LRVB_ERR_INVALID, "var[%lld] is not positive", (long long)j);
    // device scratch: X o X, the parameter vectors and nodes, (mu, v), five coefficient vectors (zero-padded past N), block sums
    DevBuf& X2 = c->mx_Xk;
    if (!c->x2_ready || X2.n < (size_t)(N * P)) {
        LRVB_TRY(buf_reserve(c, X2, (size_t)(N * P)));
        EW(square_kernel, N * P, (const double*)c->X.p, X2.p);
        c->x2_ready = true;
    }
    const i64 nblk = (N + 255) / 256, NP = N + 64;
    LRVB_TRY(buf_reserve(c, c->work1, (size_t)(2 * P + 256 + 2 * N + 5 * NP + nblk + 2 * P)));
    double* dm = c->work1.p; double* dv = dm + P; double* g = dv + P; double* mu = g + 256; double* vv = mu + N;
    double* a1 = vv + N; double* a2 = a1 + NP; double* c11 = a2 + NP; double* c12 = c11 + NP; double* c22 = c12 + NP;
    double* vpart = c22 + NP; double* gout = vpart + nblk;
    LRVB_TRY(h2d(c, dm, mean, (size_t)P));
    LRVB_TRY(h2d(c, dv, var, (size_t)P));
    LRVB_TRY(h2d(c, g, gh_x, (size_t)n_nodes));
    LRVB_TRY(h2d(c, g + 128, gh_w, (size_t)n_nodes));
    HIP_TRY(hipMemsetAsync(a1, 0, (size_t)(5 * NP) * sizeof(double), c->stream));
    LRVB_TRY(launch_gemv(c, false, N, P, 1.0, c->X.p, P, dm, 0.0, mu));
    LRVB_TRY(launch_gemv(c, false, N, P, 1.0, X2.p, P, dv, 0.0, vv));
    EW(logitnormal_coef_kernel, N, (const double*)mu, (const double*)vv, (const double*)c->y.p, (const double*)c->w.p, (const double*)g,
       (const double*)(g + 128), (int)n_nodes, a1, a2, c11, c12, c22, vpart);
    // every sum over observations of this call in ONE device buffer, [H blocks (3 P^2) | gradient (2 P) | value], summed
    // over the ranks once -- the suffix that was asked for -- before anything is copied out
    const bool want_g = grad_out != nullptr, want_H = H_blocks_out != nullptr;
    LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)(3 * P * P + 2 * P + 1)));
    double* Hb = c->Hfree.p;
    double* gred = Hb + 3 * P * P;
    double* vred = gred + 2 * P;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, c->stream, (const double*)vpart, nblk, vred);
    HIP_TRY(hipGetLastError());
    if (want_g) {
        LRVB_TRY(launch_gemv(c, true, N, P, 1.0, c->X.p, P, a1, 0.0, gred));
        LRVB_TRY(launch_gemv(c, true, N, P, 1.0, X2.p, P, a2, 0.0, gred + P));
    }
    if (want_H) {
        LRVB_TRY(weighted_tn(c, c->X.p, c->X.p, P, N, c11, Hb, c->mx_A));
        LRVB_TRY(weighted_tn(c, c->X.p, X2.p, P, N, c12, Hb + P * P, c->mx_A));
        LRVB_TRY(weighted_tn(c, X2.p, X2.p, P, N, c22, Hb + 2 * P * P, c->mx_A));
        if (!want_g) HIP_TRY(hipMemsetAsync(gred, 0, (size_t)(2 * P) * sizeof(double), c->stream));
    }
    double* first = want_H ? Hb : (want_g ? gred : vred);
    LRVB_TRY(obs_reduce(c, first, (i64)(vred + 1 - first)));
    LRVB_TRY(d2h(c, value_out, vred, 1));
    if (want_g) LRVB_TRY(d2h(c, grad_out, gred, (size_t)(2 * P)));
    if (want_H) LRVB_TRY(d2h(c, H_blocks_out, Hb, (size_t)(3 * P * P)));
    return LRVB_OK;
}

extern "C" int lrvb_dk_grad_vec(lrvb_ctx* c, const double* vec_in, int64_t V, int32_t order, const double* U,
                                const double* w_override, int32_t include_quad, double* out) {
    LRVB_TRY(ctx_bind(c));
    if (!vec_in || !out || (order > 0 && !U)) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    if (order < 0 || order > 6) LRVB_FAIL(LRVB_ERR_INVALID, "order must lie in 0..6");
    LRVB_TRY(check_len(V, c->V, "vector"));
    LRVB_TRY(data_ready(c));
    const i64 N = c->N, P = c->P;
    const bool glm = c->loss != LRVB_LOSS_NONE;
    LRVB_TRY(h2d(c, c->theta.p, vec_in, (size_t)V));
    LRVB_TRY(set_point(c, c->theta.p, false));
    // weights of this evaluation: the context's, or the caller's direction in weight space
    const double* wsrc = c->w.p;
    if (w_override && glm) {
        LRVB_TRY(reserve_obs_vec(c, c->dkw));
        LRVB_TRY(h2d(c, c->dkw.p, w_override, (size_t)N));
        wsrc = c->dkw.p;
    }
    LRVB_TRY(buf_reserve(c, c->rhs, (size_t)V));
    if (order == 0) {
        double* saved = c->w.p;                                   // the gradient pass reads c->w
        c->w.p = const_cast<double*>(wsrc);
        const int st = eval_grad_eta(c, c->stats.p, include_quad != 0);
        c->w.p = saved;
        LRVB_TRY(st);
        return d2h(c, out, c->g_eta.p, (size_t)V);
    }
    const i64 Q = order;                                          // columns: z, X u_2, ..., X u_order
    LRVB_TRY(buf_reserve(c, c->cgT, (size_t)(order * V)));
    LRVB_TRY(h2d(c, c->cgT.p, U, (size_t)(order * V)));
    if (glm) {
        // Zt (Q x P): eta's and u_2 .. u_order's GLM slices
        LRVB_TRY(buf_reserve(c, c->vtmp3, (size_t)(Q * P > V ? Q * P : V)));
        HIP_TRY(hipMemcpyAsync(c->vtmp3.p, c->eta.p + c->glm_off, (size_t)P * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        for (i64 k = 1; k < Q; ++k)
            HIP_TRY(hipMemcpyAsync(c->vtmp3.p + k * P, c->cgT.p + k * V + c->glm_off, (size_t)P * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        LRVB_TRY(buf_reserve(c, c->work1, (size_t)(N * Q)));
        if (hvp_multi_supported(c, Q) && !c->force_generic_wsyrk) {
            LRVB_TRY(reserve_obs_vec(c, c->zbuf));
            EW(fill_kernel, N, 1.0, c->zbuf.p);
            LRVB_TRY(launch_rows_times_matrix(c, 0, N, Q, c->vtmp3.p, P, c->zbuf.p, c->work1.p, Q));
        } else {
            LRVB_TRY(launch_gemm(c, false, true, N, Q, P, 1.0, c->X.p, P, c->vtmp3.p, P, 0.0, c->work1.p, Q));
        }
        LRVB_TRY(reserve_obs_vec(c, c->cw));
        EW(dk_coef_kernel, N, (int)c->loss, c->lik_info, (int)order + 1, wsrc, (const double*)c->y.p, (const double*)c->work1.p, (int)Q, c->cw.p);
    }
    // out = X^T (coef o (X u_1)), scattered into the vector layout; the quadratic term only has a second derivative
    LRVB_TRY(heta_apply_coef(c, c->cgT.p, c->rhs.p, order == 1 && include_quad != 0));
    return d2h(c, out, c->rhs.p, (size_t)V);
}

// ---- trust-region Newton-CG on the device -------------------------------------------------------------
// The optimiser the reference runs through scipy (`minimize_objective_trust_ncg`, LRVB/OptimizationUtils.py:44-75:
// scipy.optimize.minimize(method='trust-ncg') on fun_free / fun_free_grad / fun_free_hvp, or their `_cond`
// versions, LRVB/SparseObjectives.py:202-240) as one library call: the outer trust-region iteration and the
// Steihaug-Toint conjugate-gradient subproblem (Steihaug 1983; Nocedal & Wright, Algorithm 7.2, with the radius
// update of their Algorithm 4.1 as scipy parametrises it: eta = 0.15, shrink by 1/4 below rho = 1/4, double on
// the boundary above rho = 3/4) run here, every vector stays on the device, and the per-observation curvature is
// computed ONCE per accepted point -- the callback route pays a gradient pass inside every Hessian-vector call.
// Optional dense preconditioner A (D x D row-major): the iterate is y, x = A y, gradient A^T g, products A^T H A v.
namespace {
struct TrustNcg {
    lrvb_ctx* c; i64 D; const double* A;       // A on the device or nullptr
    double *y, *x, *g, *gc, *p, *z, *r, *d, *Bd, *t1, *t2, *yp;
    int nfev = 0, njev = 0, nhev = 0, nbuild = 0;
    // The products of one Steihaug-CG run share a point.  A pass over X per product is the cheaper route for a handful of them;
    // a BUILD of the point's Hessian costs about D / 86 passes (weighted SYRK on the matrix cores against an HBM-bound pass:
    // 12 at D = 1024) and makes every further product a D x D matrix-vector product (microseconds).  Ski rental: after `thr`
    // products at a point -- or at once, if the previous point needed that many -- the Hessian is built and used.
    double* Hm = nullptr; bool h_ready = false; i64 n_here = 0, prev_here = 0, thr = 0;
    int build_H() {
        LRVB_TRY(hessian_partial(c, x, true, c->stats.p));
        LRVB_TRY(stats_reduce(c));
        LRVB_TRY(hessian_finish(c, x, true, c->stats.p, Hm, D));
        h_ready = true; ++nbuild;
        return LRVB_OK;
    }
    int h_apply(const double* v, double* out) {         // H v at the point of the last eval_point
        ++n_here;
        if (Hm && !h_ready && (n_here > thr || prev_here > thr)) LRVB_TRY(build_H());
        if (h_ready) return launch_gemv(c, false, D, D, 1.0, Hm, D, v, 0.0, out);
        return hvp_apply(c, x, true, v, out);
    }

    int eval_point(const double* yv, double* f, double* gmag) {
        if (A) LRVB_TRY(launch_gemv(c, false, D, D, 1.0, A, D, yv, 0.0, x));
        else HIP_TRY(hipMemcpyAsync(x, yv, (size_t)D * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        LRVB_TRY(set_point(c, x, true));
        LRVB_TRY(eval_grad_eta(c, c->stats.p, true));
        LRVB_TRY(grad_to_free(c, x, g));
        LRVB_TRY(prepare_general_hvp(c, x));
        if (A) LRVB_TRY(launch_gemv(c, true, D, D, 1.0, A, D, g, 0.0, gc));
        LRVB_TRY(launch_dot(c, gc, gc, D, c->scal.p));
        double h[1];
        LRVB_TRY(d2h(c, f, c->stats.p, 1));
        LRVB_TRY(d2h(c, h, c->scal.p, 1));
        *gmag = sqrt(h[0]);
        ++nfev; ++njev;
        prev_here = n_here; n_here = 0; h_ready = false;       // a new point: its Hessian is not built yet
        return LRVB_OK;
    }
    int hessp(const double* v, double* out) {          // at the point of the last eval_point
        ++nhev;
        if (!A) return h_apply(v, out);
        LRVB_TRY(launch_gemv(c, false, D, D, 1.0, A, D, v, 0.0, t1));
        LRVB_TRY(h_apply(t1, t2));
        return launch_gemv(c, true, D, D, 1.0, A, D, t2, 0.0, out);
    }
    int dots(int n, const double* const* a, const double* const* b, double* host) {
        for (int k = 0; k < n; ++k) LRVB_TRY(launch_dot(c, a[k], b[k], D, c->scal.p + k));
        return d2h(c, host, c->scal.p, (size_t)n);
    }
    // model value m(q) = f + gc.q + 1/2 q.Bq  (one Hessian-vector product)
    int model(const double* q, double f, double* out) {
        LRVB_TRY(hessp(q, Bd));
        const double* a[2] = { gc, q }; const double* b[2] = { q, Bd };
        double h[2];
        LRVB_TRY(dots(2, a, b, h));
        *out = f + h[0] + 0.5 * h[1];
        return LRVB_OK;
    }
};
// roots ta <= tb of ||z + t d|| = R from zz = z.z, zd = z.d, dd = d.d (cancellation-free form)
static void boundary_roots(double zz, double zd, double dd, double R, double* ta, double* tb) {
    const double a = dd, b = 2.0 * zd, cc = zz - R * R;
    const double sq = sqrt(b * b - 4.0 * a * cc);
    const double aux = b + copysign(sq, b);
    double r1 = -aux / (2.0 * a), r2 = -2.0 * cc / aux;
    if (r1 > r2) { const double t = r1; r1 = r2; r2 = t; }
    *ta = r1; *tb = r2;
}
}  // namespace

extern "C" int lrvb_minimize_trust_ncg(lrvb_ctx* c, const double* y0, int64_t D, const double* precond,
                                       double gtol, int64_t maxiter, double initial_trust_radius,
                                       double max_trust_radius, double eta,
                                       double* y_out, double* x_out, lrvb_opt_result* res) {
    LRVB_TRY(ctx_bind(c));
    if (!y0 || !y_out || !res) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(check_len(D, c->D, "free vector"));
    LRVB_TRY(data_ready(c));
    if (!(initial_trust_radius > 0.0) || !(max_trust_radius > 0.0) || initial_trust_radius >= max_trust_radius)
        LRVB_FAIL(LRVB_ERR_INVALID, "need 0 < initial_trust_radius < max_trust_radius");
    if (!(eta >= 0.0 && eta < 0.25)) LRVB_FAIL(LRVB_ERR_INVALID, "eta must lie in [0, 0.25)");
    if (maxiter <= 0) maxiter = 200 * D;
    const size_t nA = precond ? (size_t)D * (size_t)D : 0;
    // the point's Hessian may be built inside the CG runs (TrustNcg::h_apply) where the model has a data term and the
    // resident-Hessian route is not switched off (tuning bit 3)
    // (from 256 parameters on: below, a pass and a build are both a handful of launches and the run keeps scipy's exact path)
    const bool can_build = c->loss != LRVB_LOSS_NONE && !c->no_resident && D >= 256 && D <= 8192;
    const size_t nH = can_build ? (size_t)D * (size_t)D : 0;
    LRVB_TRY(buf_reserve(c, c->opt, 12 * (size_t)D + nA + nH));
    TrustNcg o;
    o.c = c; o.D = D;
    o.Hm = can_build ? c->opt.p + 12 * (size_t)D + nA : nullptr;
    o.thr = D / 64 > 8 ? D / 64 : 8;
    double* base = c->opt.p;
    o.y = base; o.x = base + D; o.g = base + 2 * D; o.p = base + 4 * D; o.z = base + 5 * D; o.r = base + 6 * D;
    o.d = base + 7 * D; o.Bd = base + 8 * D; o.t1 = base + 9 * D; o.t2 = base + 10 * D; o.yp = base + 11 * D;
    o.A = precond ? base + 12 * D : nullptr;
    o.gc = precond ? base + 3 * D : o.g;
    if (precond) LRVB_TRY(h2d(c, base + 12 * D, precond, nA));
    LRVB_TRY(h2d(c, o.y, y0, (size_t)D));

    double f = 0.0, gmag = 0.0, radius = initial_trust_radius;
    LRVB_TRY(o.eval_point(o.y, &f, &gmag));
    int status = 0; i64 k = 0;
    while (gmag >= gtol) {
        // ---- Steihaug-Toint CG for the step p inside the ball of the current radius ---------------------------
        bool hits_boundary = false;
        const double tol_cg = fmin(0.5, sqrt(gmag)) * gmag;
        HIP_TRY(hipMemsetAsync(o.z, 0, (size_t)D * sizeof(double), c->stream));
        if (gmag < tol_cg) {
            HIP_TRY(hipMemsetAsync(o.p, 0, (size_t)D * sizeof(double), c->stream));
        } else {
            LRVB_TRY(launch_axpby(c, D, 1.0, o.gc, 0.0, o.r));            // r = gradient
            LRVB_TRY(launch_axpby(c, D, -1.0, o.gc, 0.0, o.d));           // d = -r
            double rr = gmag * gmag, zz = 0.0;
            for (;;) {
                LRVB_TRY(o.hessp(o.d, o.Bd));
                double h[3];
                LRVB_TRY(launch_dot3(c, o.d, o.Bd, o.z, o.d, o.d, o.d, D, c->scal.p));
                LRVB_TRY(d2h(c, h, c->scal.p, 3));
                const double dBd = h[0], zd = h[1], dd = h[2];
                if (dBd <= 0.0) {
                    // negative curvature: the better of the two boundary points along d
                    double ta, tb, ma, mb;
                    boundary_roots(zz, zd, dd, radius, &ta, &tb);
                    LRVB_TRY(launch_axpby(c, D, 1.0, o.z, 0.0, o.p)); LRVB_TRY(launch_axpby(c, D, ta, o.d, 1.0, o.p));
                    LRVB_TRY(launch_axpby(c, D, 1.0, o.z, 0.0, o.yp)); LRVB_TRY(launch_axpby(c, D, tb, o.d, 1.0, o.yp));
                    LRVB_TRY(o.model(o.p, f, &ma));
                    LRVB_TRY(o.model(o.yp, f, &mb));
                    if (!(ma < mb)) LRVB_TRY(launch_axpby(c, D, 1.0, o.yp, 0.0, o.p));
                    hits_boundary = true;
                    break;
                }
                const double alpha = rr / dBd;
                const double zz_next = zz + 2.0 * alpha * zd + alpha * alpha * dd;
                if (sqrt(zz_next) >= radius) {
                    double ta, tb;
                    boundary_roots(zz, zd, dd, radius, &ta, &tb);
                    LRVB_TRY(launch_axpby(c, D, 1.0, o.z, 0.0, o.p)); LRVB_TRY(launch_axpby(c, D, tb, o.d, 1.0, o.p));
                    hits_boundary = true;
                    break;
                }
                double h2[2];                                              // z += alpha d, r += alpha B d, [r.r, z.z]
                LRVB_TRY(launch_cg_update(c, D, alpha, o.d, o.Bd, o.z, o.r, c->scal.p));
                LRVB_TRY(d2h(c, h2, c->scal.p, 2));
                const double rr_next = h2[0];
                zz = h2[1];
                if (sqrt(rr_next) < tol_cg) { LRVB_TRY(launch_axpby(c, D, 1.0, o.z, 0.0, o.p)); break; }
                LRVB_TRY(launch_axpby(c, D, -1.0, o.r, rr_next / rr, o.d));   // d = -r + beta d
                rr = rr_next;
            }
        }
        // ---- ratio of actual to predicted reduction, radius update, acceptance -------------------------------
        double predicted_value;
        LRVB_TRY(o.model(o.p, f, &predicted_value));
        LRVB_TRY(launch_axpby(c, D, 1.0, o.y, 0.0, o.yp));
        LRVB_TRY(launch_axpby(c, D, 1.0, o.p, 1.0, o.yp));
        double f_new, gmag_new;
        LRVB_TRY(o.eval_point(o.yp, &f_new, &gmag_new));
        const double actual = f - f_new, predicted = f - predicted_value;
        if (!(predicted > 0.0)) {
            status = 2;                                  // the quadratic model does not decrease: stop where we are
            LRVB_TRY(o.eval_point(o.y, &f, &gmag));
            break;
        }
        const double rho = actual / predicted;
        if (rho < 0.25) radius *= 0.25;
        else if (rho > 0.75 && hits_boundary) radius = fmin(2.0 * radius, max_trust_radius);
        if (rho > eta) {
            LRVB_TRY(launch_axpby(c, D, 1.0, o.yp, 0.0, o.y));
            f = f_new; gmag = gmag_new;
        } else {
            LRVB_TRY(o.eval_point(o.y, &f, &gmag));       // rejected: restore the point state (rare)
        }
        ++k;
        if (gmag < gtol) { status = 0; break; }
        if (k >= maxiter) { status = 1; break; }
    }
    LRVB_TRY(d2h(c, y_out, o.y, (size_t)D));
    if (x_out) LRVB_TRY(d2h(c, x_out, o.x, (size_t)D));
    res->fun = f; res->jac_mag = gmag; res->trust_radius = radius;
    res->status = status; res->nit = (int32_t)k; res->nfev = o.nfev; res->njev = o.njev; res->nhev = o.nhev; res->nbuild = o.nbuild;
    return LRVB_OK;
}

// ---- blocked conjugate gradients: Q right-hand sides share every pass over the observations ------------

static int heta_apply_multi(lrvb_ctx* c, i64 Q, const double* U /* Q x V */, double* Out /* Q x V */) {
    const i64 V = c->V, P = c->P, N = c->N;
    HIP_TRY(hipMemsetAsync(Out, 0, (size_t)(Q * V) * sizeof(double), c->stream));
    if (c->loss != LRVB_LOSS_NONE) {
        const i64 Qp = Q + (Q & 1);
        const bool mfma_ok = (P % 2 == 0) && ((((uintptr_t)c->X.p) & 15) == 0) && P >= 2;
        if (hvp_multi_supported(c, 1) && !c->force_generic_wsyrk) {
            // both contractions on each row chunk while it sits in LDS: X is read once per 16 vectors
            for (i64 q0 = 0; q0 < Q; q0 += 16) {
                const i64 qn = (Q - q0 < 16) ? Q - q0 : 16;
                LRVB_TRY(launch_hvp_multi(c, qn, U + q0 * V, V, Out + q0 * V, V));
            }
        } else if (mfma_ok) {
            LRVB_TRY(launch_gemm(c, false, true, N, Q, P, 1.0, c->X.p, P, U + c->glm_off, V, 0.0, c->cgT.p, Qp));
            LRVB_TRY(launch_atb(c, c->X.p, P, c->cgT.p, Qp, N, c->cw.p, c->cgm[8].p));
            EW(scatter_rows_T_kernel, Q * P, Q, V, P, c->glm_off, c->cgm[8].p, Qp, Out);
        } else {                                  // odd row length: one fused pass per right-hand side
            LRVB_TRY(buf_reserve(c, c->vtmp3, (size_t)(V > P ? V : P)));
            for (i64 q = 0; q < Q; ++q) {
                LRVB_TRY(launch_glm_pass(c, PASS_HVP_C, nullptr, U + q * V + c->glm_off, c->vtmp3.p, nullptr, false));
                HIP_TRY(hipMemcpyAsync(Out + q * V + c->glm_off, c->vtmp3.p, (size_t)P * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
            }
        }
    }
    if (c->loss != LRVB_LOSS_NONE) LRVB_TRY(obs_reduce(c, Out, Q * V));
    if (c->quad_kind == LRVB_QUAD_DIAG) {
        EW(diag_mul_add_rows_kernel, Q * V, V, c->quad_scale, c->quadA.p, U, Out);
    } else if (c->quad_kind == LRVB_QUAD_DENSE) {
        LRVB_TRY(launch_gemm(c, false, true, Q, V, V, c->quad_scale, U, V, c->quadA.p, V, 1.0, Out, V));
    }
    return LRVB_OK;
}

// Out (Q x D) = H_free Vb^T row by row.  Requires set_point + eval_grad_eta (+ prepare_general_hvp) done.
static int hvp_apply_multi(lrvb_ctx* c, i64 Q, const double* Vb, double* Out) {
    const i64 D = c->D, V = c->V;
    double* U = c->cgm[6].p; double* W = c->cgm[7].p;
    if (c->all_box) {
        EW(mul_rows_kernel, Q * D, D, c->j1.p, Vb, U);
        LRVB_TRY(heta_apply_multi(c, Q, U, W));
        EW(fma3_rows_kernel, Q * D, D, c->g_eta.p, c->j2.p, Vb, c->j1.p, W, Out);
        return LRVB_OK;
    }
    LRVB_TRY(launch_gemm(c, false, true, Q, V, D, 1.0, Vb, D, c->Jdense.p, D, 0.0, U, V));       // U = Vb J^T
    LRVB_TRY(heta_apply_multi(c, Q, U, W));
    LRVB_TRY(launch_gemm(c, false, false, Q, D, V, 1.0, W, V, c->Jdense.p, D, 0.0, Out, D));     // W J
    return launch_gemm(c, false, false, Q, D, D, 1.0, Vb, D, c->Tdense.p, D, 1.0, Out, D);        // + Vb T (T symmetric)
}


// ---- blocked CG, fused form (box layouts, no preconditioner): an iteration is FIVE launches and no host round trip ------

static bool cg_multi_fused_ok(const lrvb_ctx* c, const double* Minv, i64 Q) {
    return Q <= 400 && c->all_box && !Minv && c->loss != LRVB_LOSS_NONE && c->quad_kind != LRVB_QUAD_DENSE && c->V == c->D &&
           hvp_multi_supported(c, 1) && !c->force_generic_wsyrk;
}

// the loop of lrvb_cg_solve_multi after the common set-up (Bd, Xd, Rd, Pd = 0 in place; s[0..Q) = |b|^2 on the device)
static int cg_multi_fused_loop(lrvb_ctx* c, i64 Q, i64 D, double tol, i64 maxiter, double* Xd, double* Rd, double* Pd,
                               std::vector<int>& info, std::vector<int64_t>& iters, bool resident) {
    double* s = c->scal.p;
    double* U = c->cgm[6].p; double* W = c->cgm[7].p;
    LRVB_TRY(ensure_aux(c));
    LRVB_TRY(pinned_reserve(c, 4096));
    // live = (|b| > 0), rho_prev = 0, iterations = 0
    std::vector<double> init((size_t)(5 * Q), 0.0), hb((size_t)Q);
    LRVB_TRY(d2h(c, hb.data(), s, (size_t)Q));
    for (i64 q = 0; q < Q; ++q) { init[q] = hb[q]; init[3 * Q + q] = hb[q] > 0.0 ? 1.0 : 0.0; if (hb[q] == 0.0) HIP_TRY(hipMemsetAsync(Xd + q * D, 0, (size_t)D * sizeof(double), c->stream)); }
    LRVB_TRY(h2d(c, s, init.data(), (size_t)(5 * Q)));
    const double* quadA = (!resident && c->quad_kind == LRVB_QUAD_DIAG) ? c->quadA.p : nullptr;
    const double* j1 = resident ? nullptr : c->j1.p;
    double* status = c->host_pinned + 2048;                        // [live (Q)] of the iteration whose head kernel ran last
    auto queue_iteration = [&](i64 it) -> int {
        hipLaunchKernelGGL(cg_multi_head_kernel, dim3((unsigned)Q), dim3(256), 0, c->stream, D, it, tol, j1,
                           (const double*)Rd, Pd, U, s, Q);
        HIP_TRY(hipGetLastError());
        // status of THIS iteration's test, copied on the side stream as soon as the head kernel is done
        HIP_TRY(hipEventRecord(c->aux_ev[0], c->stream));
        HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->aux_ev[0], 0));
        // the per-parity snapshot, not the live flags: head(it + 2) -- the next writer of this slot -- is queued only after the host has
        // consumed this copy, so every rank reads the same flags for iteration `it` and queues the same number of reductions
        HIP_TRY(hipMemcpyAsync(status + (it & 1) * 1024, s + (5 + (it & 1)) * Q, (size_t)Q * sizeof(double), hipMemcpyDeviceToHost, c->aux_stream));
        if (resident) {
            // W = U H: the whole block against the resident matrix (H symmetric), identical on every rank -- no reduction
            LRVB_TRY(launch_symm_block(c, Q, D, U, c->Hres.p, W));
        } else {
        HIP_TRY(hipMemsetAsync(W, 0, (size_t)(Q * D) * sizeof(double), c->stream));
        for (i64 q0 = 0; q0 < Q; q0 += 16) {
            const i64 qn = (Q - q0 < 16) ? Q - q0 : 16;
            c->hm_live = s + 3 * Q + q0;
            const int st = launch_hvp_multi(c, qn, U + q0 * D, D, W + q0 * D, D);
            c->hm_live = nullptr;
            LRVB_TRY(st);
        }
        LRVB_TRY(obs_reduce(c, W, Q * D));
        }                            // every rank queues the same iterations: the stop decision below
                                                                      // reads per-iteration snapshots of reduced (rank-identical) scalars
        hipLaunchKernelGGL(cg_multi_tail_kernel, dim3((unsigned)Q), dim3(256), 0, c->stream, D, c->quad_scale, quadA, j1,
                           (const double*)c->j2.p, (const double*)c->g_eta.p, (const double*)W, (const double*)U, (const double*)Pd, Xd, Rd,
                           (const double*)s, Q);
        HIP_TRY(hipGetLastError());
        return LRVB_OK;
    };
    i64 queued = 0;
    bool stop = false;
    if (maxiter > 0) { LRVB_TRY(queue_iteration(0)); queued = 1; }
    for (i64 it = 0; it < maxiter && !stop; ++it) {
        if (it + 1 < maxiter) { LRVB_TRY(queue_iteration(it + 1)); queued = it + 2; }      // one ahead of the test below
        // wait for the status copy of iteration `it` only (the side stream), not for the main stream
        HIP_TRY(hipStreamSynchronize(c->aux_stream));
        // the copy of iteration it + 1 may be queued behind it on the side stream: wait covers both, which is fine (it only
        // means the head kernel of it + 1 ran, i.e. iteration `it` is complete)
        const double* st = status + (it & 1) * 1024;
        bool any = false;
        for (i64 q = 0; q < Q; ++q) any = any || (st[q] != 0.0);
        if (!any) stop = true;
    }
    (void)queued;
    std::vector<double> fin((size_t)(5 * Q));
    LRVB_TRY(d2h(c, fin.data(), s, (size_t)(5 * Q)));
    for (i64 q = 0; q < Q; ++q) {
        info[(size_t)q] = fin[3 * Q + q] != 0.0 ? (int)maxiter : 0;
        iters[(size_t)q] = (int64_t)fin[4 * Q + q];
    }
    return LRVB_OK;
}

extern "C" int lrvb_cg_solve_multi(lrvb_ctx* c, const double* free_in, const double* B, const double* X0,
                                   const double* Minv, double tol, int64_t maxiter, int64_t D, int64_t Q,
                                   double* X_out, int* info_out, int64_t* iters_out) {
    // Same point as the previous product / solve on this context, nothing else in between: eta, the packing Jacobian,
    // d f / d eta and the per-observation curvature are still in place (the reference's ConjugateGradientSolver is built
    // for ONE point x0 and solves for many right-hand sides there, LRVB/ConjugateGradient.py:63-105) -- no second
    // gradient pass over X.
    const bool reuse = same_point(c, free_in, D, true);
    const bool prepared = reuse && c->hvp_pt_prepared;
    LRVB_TRY(ctx_bind(c));
    if (!free_in || !B || !X_out || Q <= 0) LRVB_FAIL(LRVB_ERR_INVALID, "bad argument");
    LRVB_TRY(check_len(D, c->D, "free vector"));
    LRVB_TRY(data_ready(c));
    if (maxiter <= 0) maxiter = 10 * D;
    const i64 V = c->V, Qp = Q + (Q & 1);
    const size_t qd = (size_t)Q * (size_t)D, qv = (size_t)Q * (size_t)V;
    for (int k = 0; k < 6; ++k) LRVB_TRY(buf_reserve(c, c->cgm[k], qd));
    LRVB_TRY(buf_reserve(c, c->cgm[6], qv));
    LRVB_TRY(buf_reserve(c, c->cgm[7], qv));
    LRVB_TRY(buf_reserve(c, c->cgm[8], (size_t)((c->P > 0 ? c->P : 1) * Qp)));
    bool resident = false;                                 // the Hessian of this point is resident: block products are Q x D x D GEMMs
    LRVB_TRY(hres_matches(c, free_in, D, &resident));
    if (c->loss != LRVB_LOSS_NONE && !resident) {
        LRVB_TRY(buf_reserve(c, c->cgT, (size_t)(c->N * Qp)));
        HIP_TRY(hipMemsetAsync(c->cgT.p, 0, (size_t)(c->N * Qp) * sizeof(double), c->stream));     // keeps the padding column zero
    }
    double *Bd = c->cgm[0].p, *Xd = c->cgm[1].p, *Rd = c->cgm[2].p, *Pd = c->cgm[3].p, *Qd = c->cgm[4].p, *Zd = c->cgm[5].p;
    if (Minv) { LRVB_TRY(buf_reserve(c, c->Hfree, (size_t)D * (size_t)D)); LRVB_TRY(h2d(c, c->Hfree.p, Minv, (size_t)D * (size_t)D)); }
    LRVB_TRY(h2d(c, Bd, B, qd));
    if (!resident) {
        if (!reuse) {
            LRVB_TRY(h2d(c, c->theta.p, free_in, (size_t)D));
            LRVB_TRY(set_point(c, c->theta.p, true));
            LRVB_TRY(eval_grad_eta(c, c->stats.p, true));
        }
        if (!prepared) LRVB_TRY(prepare_general_hvp(c, c->theta.p));
    }
    auto block_product = [&](const double* Vb, double* Out) -> int {
        return resident ? launch_symm_block(c, Q, D, Vb, c->Hres.p, Out) : hvp_apply_multi(c, Q, Vb, Out);
    };
    // scalars: s[0..Q) = ||b||^2 | rr | rz | pq | alpha | beta | minus_alpha | one
    LRVB_TRY(buf_reserve(c, c->scal, (size_t)(8 * Q + 16)));
    double* s = c->scal.p;
    std::vector<double> hs((size_t)(4 * Q)), coef((size_t)(4 * Q));
    std::vector<double> bnorm((size_t)Q), rho_prev((size_t)Q, 0.0);
    std::vector<int> info((size_t)Q, 0); std::vector<int64_t> iters((size_t)Q, 0); std::vector<char> active((size_t)Q, 1);
    hipLaunchKernelGGL(rows_dot_kernel, dim3((unsigned)Q), dim3(256), 0, c->stream, Q, D, Bd, Bd, s);
    HIP_TRY(hipGetLastError());
    if (X0) {
        LRVB_TRY(h2d(c, Xd, X0, qd));
        LRVB_TRY(block_product(Xd, Qd));
        HIP_TRY(hipMemcpyAsync(Rd, Bd, qd * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
        LRVB_TRY(launch_axpby(c, (i64)qd, -1.0, Qd, 1.0, Rd));
    } else {
        HIP_TRY(hipMemsetAsync(Xd, 0, qd * sizeof(double), c->stream));
        HIP_TRY(hipMemcpyAsync(Rd, Bd, qd * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    HIP_TRY(hipMemsetAsync(Pd, 0, qd * sizeof(double), c->stream));
    LRVB_TRY(d2h(c, hs.data(), s, (size_t)Q));
    i64 n_active = 0;
    for (i64 q = 0; q < Q; ++q) {
        bnorm[q] = sqrt(hs[q]);
        if (bnorm[q] == 0.0) { active[q] = 0; HIP_TRY(hipMemsetAsync(Xd + q * D, 0, (size_t)D * sizeof(double), c->stream)); }
        else { info[q] = (int)maxiter; ++n_active; }
    }
    if ((resident && !Minv && Q <= 400) || cg_multi_fused_ok(c, Minv, Q)) {
        LRVB_TRY(cg_multi_fused_loop(c, Q, D, tol, maxiter, Xd, Rd, Pd, info, iters, resident));
        LRVB_TRY(d2h(c, X_out, Xd, qd));
        for (i64 q = 0; q < Q; ++q) { if (info_out) info_out[q] = info[q]; if (iters_out) iters_out[q] = iters[q]; }
        if (!resident) remember_point(c, free_in, D, true, true);
        return LRVB_OK;
    }
    for (i64 it = 0; it < maxiter && n_active > 0; ++it) {
        if (Minv) LRVB_TRY(launch_gemm(c, false, true, Q, D, D, 1.0, Rd, D, c->Hfree.p, D, 0.0, Zd, D));   // Z = R Minv^T
        const double* Z = Minv ? Zd : Rd;
        hipLaunchKernelGGL(rows_dot_kernel, dim3((unsigned)Q), dim3(256), 0, c->stream, Q, D, Rd, Rd, s + Q);
        hipLaunchKernelGGL(rows_dot_kernel, dim3((unsigned)Q), dim3(256), 0, c->stream, Q, D, Rd, Z, s + 2 * Q);
        HIP_TRY(hipGetLastError());
        LRVB_TRY(d2h(c, hs.data(), s + Q, (size_t)(2 * Q)));
        for (i64 q = 0; q < Q; ++q) {
            if (!active[q]) { coef[q] = 0.0; coef[Q + q] = 1.0; continue; }          // p frozen
            if (sqrt(hs[q]) < tol * bnorm[q]) { active[q] = 0; info[q] = 0; iters[q] = it; --n_active; coef[q] = 0.0; coef[Q + q] = 1.0; continue; }
            const double rho = hs[Q + q];
            coef[q] = 1.0;                                    // p = 1 * z + beta p
            coef[Q + q] = (it > 0 && rho_prev[q] != 0.0) ? rho / rho_prev[q] : 0.0;
            rho_prev[q] = rho;
        }
        if (n_active == 0) break;
        LRVB_TRY(h2d(c, s + 4 * Q, coef.data(), (size_t)(2 * Q)));
        EW(rows_axpby_kernel, (i64)qd, D, s + 4 * Q, Z, s + 5 * Q, Pd);
        LRVB_TRY(block_product(Pd, Qd));
        hipLaunchKernelGGL(rows_dot_kernel, dim3((unsigned)Q), dim3(256), 0, c->stream, Q, D, Pd, Qd, s + 3 * Q);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(cg_multi_alpha_kernel, dim3((unsigned)((Q + 63) / 64)), dim3(64), 0, c->stream, (int)Q, s);
        HIP_TRY(hipGetLastError());
        for (i64 q = 0; q < Q; ++q) if (active[q]) iters[q] = it + 1;
        EW(rows_axpby_kernel, (i64)qd, D, s + 4 * Q, Pd, s + 5 * Q, Xd);          // x += alpha p
        EW(rows_axpby_kernel, (i64)qd, D, s + 6 * Q, Qd, s + 7 * Q, Rd);          // r -= alpha q
    }
    LRVB_TRY(d2h(c, X_out, Xd, qd));
    for (i64 q = 0; q < Q; ++q) { if (info_out) info_out[q] = info[q]; if (iters_out) iters_out[q] = iters[q]; }
    if (!resident) remember_point(c, free_in, D, true, true);
    return LRVB_OK;
}

// ---- profiling ----------------------------------------------------------------------------------
extern "C" int lrvb_profile_enable(lrvb_ctx* c, int on) {
    if (!c) LRVB_FAIL(LRVB_ERR_INVALID, "null context");
    c->prof_on = on != 0;
    return LRVB_OK;
}
int prof_mark(lrvb_ctx* c, int which) {
    std::vector<hipEvent_t>& pool = c->ev_pool[which];
    if (c->ev_used[which] == pool.size()) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreate(&e));
        pool.push_back(e);
    }
    HIP_TRY(hipEventRecord(pool[c->ev_used[which]++], c->stream));
    return LRVB_OK;
}
static int prof_collect(lrvb_ctx* c) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < PROF_POOLS; ++k) {
        double ms_sum = 0.0; int64_t calls = 0;
        for (size_t i = 0; i + 1 < c->ev_used[k]; i += 2) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, c->ev_pool[k][i], c->ev_pool[k][i + 1]));
            ms_sum += ms; ++calls;
        }
        c->ev_used[k] = 0;
        if (k == PROF_WSYRK) { c->prof.wsyrk_ms += ms_sum; c->prof.wsyrk_calls += calls; }
        if (k == PROF_PASS)  { c->prof.pass_ms  += ms_sum; c->prof.pass_calls  += calls; }
        if (k == PROF_BUILD) { c->prof.build_ms += ms_sum; c->prof.build_calls += calls; }
        if (k == PROF_REDUCE) { c->prof.reduce_ms += ms_sum; c->prof.reduce_calls += calls; }
    }
    return LRVB_OK;
}
extern "C" int lrvb_profile_get(lrvb_ctx* c, lrvb_prof* out) {
    if (!c || !out) LRVB_FAIL(LRVB_ERR_INVALID, "null argument");
    LRVB_TRY(ctx_bind(c));
    LRVB_TRY(prof_collect(c));
    *out = c->prof;
    return LRVB_OK;
}
extern "C" int lrvb_profile_reset(lrvb_ctx* c) {
    if (!c) LRVB_FAIL(LRVB_ERR_INVALID, "null context");
    LRVB_TRY(ctx_bind(c));
    LRVB_TRY(prof_collect(c));
    c->prof = lrvb_prof{};
    return LRVB_OK;
}
