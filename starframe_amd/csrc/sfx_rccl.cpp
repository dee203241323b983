// sfx_rccl.cpp -- libsfx_rccl.so: the slab halo exchange over plain RCCL (include/sfx_rccl.h).
// Host code only: it moves the device buffers libsfx owns with ncclSend / ncclRecv on the world's stream.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>

#include "../../include/sfx_rccl.h"

static_assert(SFX_RCCL_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");

struct sfx_rccl_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1, device = 0;
    double* stage = nullptr;   // device staging of the stats reduction (64 doubles)
};

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define NCCLCHK(expr)                                                                                          \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(SFX_ERR_DEVICE, "%s failed: %s", #expr, ncclGetErrorString(r_));    \
    } while (0)
#define HIPCHK(expr)                                                                                           \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(SFX_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

constexpr uint32_t kStage = 64;

}  // namespace

extern "C" {

const char* sfx_rccl_last_error(void) { return g_err.c_str(); }

int sfx_rccl_unique_id(unsigned char id[SFX_RCCL_ID_BYTES]) {
    if (!id) return fail(SFX_ERR_INVALID, "null id");
    ncclUniqueId u;
    NCCLCHK(ncclGetUniqueId(&u));
    memcpy(id, u.internal, SFX_RCCL_ID_BYTES);
    return SFX_OK;
}

int sfx_rccl_comm_create(const unsigned char id[SFX_RCCL_ID_BYTES], int rank, int nranks, int device, sfx_rccl_comm** out) {
    if (!id || !out) return fail(SFX_ERR_INVALID, "null argument");
    *out = nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(SFX_ERR_INVALID, "rank %d of %d", rank, nranks);
    sfx_rccl_comm* c = new (std::nothrow) sfx_rccl_comm();
    if (!c) return fail(SFX_ERR_NOMEM, "out of host memory");
    c->rank = rank; c->nranks = nranks; c->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&c->stage), kStage * sizeof(double));
    if (e != hipSuccess) { delete c; return fail(SFX_ERR_DEVICE, "device %d set-up failed: %s", device, hipGetErrorString(e)); }
    ncclUniqueId u;
    memcpy(u.internal, id, SFX_RCCL_ID_BYTES);
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        (void)hipFree(c->stage);
        delete c;
        return fail(SFX_ERR_DEVICE, "ncclCommInitRank failed: %s", ncclGetErrorString(r));
    }
    *out = c;
    return SFX_OK;
}

void sfx_rccl_comm_abort(sfx_rccl_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)ncclCommAbort(c->comm);   // does not wait for the peers: outstanding operations are dropped
    if (c->stage) (void)hipFree(c->stage);
    delete c;
}

void sfx_rccl_comm_destroy(sfx_rccl_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->stage) (void)hipFree(c->stage);
    delete c;
}

int sfx_rccl_halo_exchange(sfx_rccl_comm* c, sfx_world* w, int left, int right, uint32_t n_send_left,
                           uint32_t n_send_right, uint32_t n_recv_left, uint32_t n_recv_right) {
    if (!c || !w) return fail(SFX_ERR_INVALID, "null argument");
    if ((left >= c->nranks) || (right >= c->nranks)) return fail(SFX_ERR_INVALID, "neighbour rank out of range");
    if (left < 0 && (n_send_left || n_recv_left)) return fail(SFX_ERR_INVALID, "left halo without a left neighbour");
    if (right < 0 && (n_send_right || n_recv_right)) return fail(SFX_ERR_INVALID, "right halo without a right neighbour");
    // one peer as BOTH neighbours pairs the blocks up in issue order -- its left block would land where its right block
    // belongs: only a rank that names ITSELF (the single-GPU self test) may do that
    if (left >= 0 && left == right && left != c->rank)
        return fail(SFX_ERR_UNSUPPORTED, "rank %d as both the left and the right neighbour (a ring of two slabs) is not supported", left);
    uint32_t n_send = 0, n_recv = 0;
    int rc = sfx_halo_counts(w, &n_send, &n_recv);
    if (rc) return fail(rc, "%s", sfx_last_error(w));
    if (n_send_left + n_send_right != n_send || n_recv_left + n_recv_right != n_recv)
        return fail(SFX_ERR_INVALID, "halo counts %u + %u / %u + %u do not match sfx_halo_set's %u / %u", n_send_left, n_send_right,
                    n_recv_left, n_recv_right, n_send, n_recv);
    void* stream = nullptr;
    float *send = nullptr, *recv = nullptr;
    rc = sfx_halo_stream(w, &stream);   // the world's stream, or the halo stream of a world with the exchange overlapped
    if (!rc) rc = sfx_halo_buffers(w, &send, &recv);
    if (!rc) rc = sfx_halo_pack(w);
    if (rc) return fail(rc, "%s", sfx_last_error(w));
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    {   // RCCL work must go on a stream of the communicator's device
        hipDevice_t sd = 0;
        if (st && hipStreamGetDevice(st, &sd) == hipSuccess && (int)sd != c->device)
            return fail(SFX_ERR_INVALID, "the world runs on device %d, the communicator was made for device %d", (int)sd, c->device);
    }
    // every message is a run of 8-float body records; the two directions of one neighbour pair are matched by the
    // order of the calls inside the group (sends and receives to / from one peer pair up in issue order)
    const size_t sl = 8 * (size_t)n_send_left, sr = 8 * (size_t)n_send_right, rl = 8 * (size_t)n_recv_left, rr = 8 * (size_t)n_recv_right;
    if (sl + sr + rl + rr) {
        // an error inside the group must not leave it open (every later call on the communicator would queue into it):
        // the first error is kept, ncclGroupEnd is always called
        ncclResult_t first = ncclSuccess;
        auto keep = [&](ncclResult_t r) { if (first == ncclSuccess && r != ncclSuccess) first = r; };
        NCCLCHK(ncclGroupStart());
        if (left >= 0 && sl) keep(ncclSend(send, sl, ncclFloat, left, c->comm, st));
        if (right >= 0 && sr) keep(ncclSend(send + sl, sr, ncclFloat, right, c->comm, st));
        // what comes from the left neighbour is ITS right block, from the right neighbour its left block
        if (left >= 0 && rl) keep(ncclRecv(recv, rl, ncclFloat, left, c->comm, st));
        if (right >= 0 && rr) keep(ncclRecv(recv + rl, rr, ncclFloat, right, c->comm, st));
        keep(ncclGroupEnd());
        if (first != ncclSuccess) return fail(SFX_ERR_DEVICE, "halo exchange failed: %s", ncclGetErrorString(first));
    }
    rc = sfx_halo_unpack(w);
    if (rc) return fail(rc, "%s", sfx_last_error(w));
    return SFX_OK;
}

int sfx_rccl_sendrecv(sfx_rccl_comm* c, void* hip_stream, const float* send_dev, size_t n_send, int to, float* recv_dev,
                      size_t n_recv, int from) {
    if (!c || (n_send && !send_dev) || (n_recv && !recv_dev)) return fail(SFX_ERR_INVALID, "null argument");
    if (to >= c->nranks || from >= c->nranks) return fail(SFX_ERR_INVALID, "peer rank out of range");
    HIPCHK(hipSetDevice(c->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
    ncclResult_t first = ncclSuccess;
    auto keep = [&](ncclResult_t r) { if (first == ncclSuccess && r != ncclSuccess) first = r; };
    NCCLCHK(ncclGroupStart());
    if (to >= 0 && n_send) keep(ncclSend(send_dev, n_send, ncclFloat, to, c->comm, st));
    if (from >= 0 && n_recv) keep(ncclRecv(recv_dev, n_recv, ncclFloat, from, c->comm, st));
    keep(ncclGroupEnd());   // (always: an open group would swallow every later call)
    if (first != ncclSuccess) return fail(SFX_ERR_DEVICE, "send / receive failed: %s", ncclGetErrorString(first));
    return SFX_OK;
}

int sfx_rccl_allreduce_stats(sfx_rccl_comm* c, double* sums, uint32_t n_sum, double* maxs, uint32_t n_max) {
    if (!c || (n_sum && !sums) || (n_max && !maxs)) return fail(SFX_ERR_INVALID, "null argument");
    if (n_sum + n_max > kStage) return fail(SFX_ERR_UNSUPPORTED, "at most %u figures", kStage);
    HIPCHK(hipSetDevice(c->device));
    if (n_sum) HIPCHK(hipMemcpy(c->stage, sums, n_sum * sizeof(double), hipMemcpyHostToDevice));
    if (n_max) HIPCHK(hipMemcpy(c->stage + n_sum, maxs, n_max * sizeof(double), hipMemcpyHostToDevice));
    if (n_sum) NCCLCHK(ncclAllReduce(c->stage, c->stage, n_sum, ncclDouble, ncclSum, c->comm, nullptr));
    if (n_max) NCCLCHK(ncclAllReduce(c->stage + n_sum, c->stage + n_sum, n_max, ncclDouble, ncclMax, c->comm, nullptr));
    HIPCHK(hipStreamSynchronize(nullptr));
    if (n_sum) HIPCHK(hipMemcpy(sums, c->stage, n_sum * sizeof(double), hipMemcpyDeviceToHost));
    if (n_max) HIPCHK(hipMemcpy(maxs, c->stage + n_sum, n_max * sizeof(double), hipMemcpyDeviceToHost));
    return SFX_OK;
}

}  // extern "C"
