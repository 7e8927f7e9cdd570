/*
 * ddc_multi.cpp -- the multi-GPU part of the thin C ABI (include/perseus_ddc.h,
 * "multi-GPU" section): RCCL over xGMI, called directly (librccl), no torch.
 *
 * The stream shards as independent receivers -- the reference models up to 8 of
 * them as 8 descriptors with one transfer queue each (perseus-sdr.c:43-47,
 * perseus-in.h:87) -- so the data path needs no collective.  What does cross
 * GPUs is
 *   - the configuration (stage plan, taps, NCO word): a few KB, root -> all
 *     (ncclBroadcast), once per change;
 *   - optionally the decimated output of every GPU, gathered on a root GPU
 *     (BASELINE config 4): grouped ncclSend / ncclRecv peer -> root, so that all
 *     of the root's xGMI links carry one peer each (xGMI is point to point);
 *     issued on a side stream so the transfer of batch k runs under the kernels
 *     of batch k+1.
 * Two ways to own the communicators, same calls afterwards:
 *   pddc_comm_init_rank  one process per GPU (bench.py under torch.distributed.run:
 *                        rank 0 makes the id, the launcher's store carries it)
 *   pddc_comm_init_all   one process driving several GPUs (a C host with several
 *                        perseus_descr, one per GPU); collective calls on all of
 *                        its communicators go between pddc_comm_group_start/_end.
 */
#include "../../include/perseus_ddc.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

extern "C" __attribute__((visibility("hidden"))) int pddc_set_error_(int code, const char *fmt, ...);   /* ddc_pipeline.cpp */

#define HIP_TRYM(expr)                                                                             \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return pddc_set_error_(e__ == hipErrorOutOfMemory ? PDDC_ENOMEM : PDDC_EHIP, "%s: %s", #expr, \
                                   hipGetErrorString(e__));                                        \
    } while (0)

#define NCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        ncclResult_t r__ = (expr);                                                                 \
        if (r__ != ncclSuccess)                                                                    \
            return pddc_set_error_(PDDC_ECOMM, "%s: %s", #expr, ncclGetErrorString(r__));          \
    } while (0)

struct pddc_comm {
    ncclComm_t comm = nullptr;
    int nranks = 0, rank = 0, device = 0;
    hipStream_t side = nullptr;          /* gathers run here, beside the compute stream      */
    hipEvent_t ev_ready = nullptr;       /* compute stream -> side stream: the batch exists  */
    hipEvent_t ev_done[2] = { nullptr, nullptr };   /* side stream: transfer n has left / arrived (n & 1) */
    unsigned long long n_gathers = 0;    /* side-stream gathers started so far                */
    void *d_scratch = nullptr;           /* bounce buffer of the host-level helpers          */
    size_t scratch_cap = 0;
};

/* pddc_comm_group_start/_end bracket the per-communicator calls of ONE collective when a single
 * thread drives several GPUs.  RCCL only enqueues the grouped operations at the outermost
 * ncclGroupEnd, so the "transfer done" event of a side-stream gather cannot be recorded where the
 * gather is called: it is recorded here, after the group has ended.                              */
static thread_local int g_group_depth = 0;
static thread_local std::vector<pddc_comm *> g_group_pending;

static int comm_finish_init(pddc_comm *c)
{
    HIP_TRYM(hipSetDevice(c->device));
    HIP_TRYM(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    HIP_TRYM(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    HIP_TRYM(hipEventCreateWithFlags(&c->ev_done[0], hipEventDisableTiming));
    HIP_TRYM(hipEventCreateWithFlags(&c->ev_done[1], hipEventDisableTiming));
    return PDDC_OK;
}

static int ensure_scratch(pddc_comm *c, size_t nbytes)
{
    if (c->scratch_cap >= nbytes)
        return PDDC_OK;
    if (c->d_scratch)
        HIP_TRYM(hipFree(c->d_scratch));
    c->d_scratch = nullptr;
    c->scratch_cap = 0;
    const size_t cap = nbytes < 4096 ? 4096 : nbytes;
    HIP_TRYM(hipMalloc(&c->d_scratch, cap));
    c->scratch_cap = cap;
    return PDDC_OK;
}

/* Two librccl are on these boxes: the one this library was linked against (/opt/rocm/lib) and, in a Python process that
 * has imported torch, torch/lib/librccl.so, which the dynamic loader may have bound first.  The RUNNING library's version
 * is compared with the header this file was compiled against before the first communicator is made: a different MAJOR
 * version means different struct layouts and wire protocol -- fail here, loudly, not inside the first collective on 8 GPUs. */
static int check_rccl_version(void)
{
    static int checked = 0;                 /* 0 not yet, 1 fine, -1 refused */
    static int running = 0;
    if (checked == 0) {
        if (ncclGetVersion(&running) != ncclSuccess)
            running = -1;
        const int run_major = running >= 10000 ? running / 10000 : running / 1000;
        const int hdr_major = NCCL_VERSION_CODE >= 10000 ? NCCL_VERSION_CODE / 10000 : NCCL_VERSION_CODE / 1000;
        checked = (running > 0 && run_major == hdr_major) ? 1 : -1;
        if (getenv("PDDC_DEBUG") || checked < 0)
            fprintf(stderr, "[pddc] RCCL: running library version code %d, compiled against %d (%d.%d.%d)%s\n", running,
                    (int)NCCL_VERSION_CODE, NCCL_MAJOR, NCCL_MINOR, NCCL_PATCH, checked < 0 ? " -- MAJOR VERSION MISMATCH" : "");
    }
    if (checked < 0)
        return pddc_set_error_(PDDC_ECOMM, "the RCCL library bound at run time (version code %d) does not match the one this "
                               "library was built against (%d): two librccl on the path?", running, (int)NCCL_VERSION_CODE);
    return PDDC_OK;
}

extern "C" {

/* version codes of the RCCL library in use and of the header compiled against (either pointer may be NULL) */
int pddc_comm_rccl_version(int *running, int *compiled)
{
    int v = 0;
    if (ncclGetVersion(&v) != ncclSuccess)
        return pddc_set_error_(PDDC_ECOMM, "ncclGetVersion failed");
    if (running)
        *running = v;
    if (compiled)
        *compiled = (int)NCCL_VERSION_CODE;
    return PDDC_OK;
}

int pddc_comm_get_unique_id(void *id)
{
    if (!id)
        return pddc_set_error_(PDDC_EINVAL, "null id");
    static_assert(sizeof(ncclUniqueId) == PDDC_COMM_ID_BYTES, "unique id size");
    {
        int rc = check_rccl_version();
        if (rc)
            return rc;
    }
    ncclUniqueId u;
    NCCL_TRY(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return PDDC_OK;
}

int pddc_comm_init_rank(pddc_comm **out, int nranks, int rank, const void *id, int device)
{
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks)
        return pddc_set_error_(PDDC_EINVAL, "bad communicator arguments (nranks %d rank %d)", nranks, rank);
    *out = nullptr;
    int ndev = pddc_device_count();
    if (ndev <= 0)
        return pddc_set_error_(PDDC_ENODEV, "no HIP device visible (no CPU fallback)");
    if (device < 0 || device >= ndev)
        return pddc_set_error_(PDDC_ENODEV, "device %d out of range (0..%d)", device, ndev - 1);
    {
        int rc = check_rccl_version();
        if (rc)
            return rc;
    }
    HIP_TRYM(hipSetDevice(device));
    pddc_comm *c = new (std::nothrow) pddc_comm();
    if (!c)
        return pddc_set_error_(PDDC_ENOMEM, "out of host memory");
    c->nranks = nranks;
    c->rank = rank;
    c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
    if (r != ncclSuccess) {
        delete c;
        return pddc_set_error_(PDDC_ECOMM, "ncclCommInitRank(%d of %d): %s", rank, nranks, ncclGetErrorString(r));
    }
    int rc = comm_finish_init(c);
    if (rc) {
        pddc_comm_destroy(c);
        return rc;
    }
    *out = c;
    return PDDC_OK;
}

int pddc_comm_init_all(pddc_comm **comms, int ndev, const int *devices)
{
    if (!comms || ndev < 1 || ndev > 64)
        return pddc_set_error_(PDDC_EINVAL, "bad communicator arguments (ndev %d)", ndev);
    for (int i = 0; i < ndev; ++i)
        comms[i] = nullptr;
    const int have = pddc_device_count();
    if (have <= 0)
        return pddc_set_error_(PDDC_ENODEV, "no HIP device visible (no CPU fallback)");
    std::vector<int> devs(ndev);
    for (int i = 0; i < ndev; ++i) {
        devs[i] = devices ? devices[i] : i;
        if (devs[i] < 0 || devs[i] >= have)
            return pddc_set_error_(PDDC_ENODEV, "device %d out of range (0..%d)", devs[i], have - 1);
        for (int j = 0; j < i; ++j)
            if (devs[j] == devs[i])     /* RCCL refuses two ranks on one GPU */
                return pddc_set_error_(PDDC_EINVAL, "device %d listed twice: one communicator rank per GPU", devs[i]);
    }
    std::vector<ncclComm_t> raw(ndev, nullptr);
    {
        int rc = check_rccl_version();
        if (rc)
            return rc;
    }
    NCCL_TRY(ncclCommInitAll(raw.data(), ndev, devs.data()));
    for (int i = 0; i < ndev; ++i) {
        pddc_comm *c = new (std::nothrow) pddc_comm();
        int rc = c ? PDDC_OK : pddc_set_error_(PDDC_ENOMEM, "out of host memory");
        if (!rc) {
            c->comm = raw[i];
            raw[i] = nullptr;
            c->nranks = ndev;
            c->rank = i;
            c->device = devs[i];
            comms[i] = c;
            rc = comm_finish_init(c);
        }
        if (rc) {
            for (int j = 0; j < ndev; ++j) {
                if (comms[j])
                    pddc_comm_destroy(comms[j]);
                else if (raw[j])
                    ncclCommDestroy(raw[j]);
                comms[j] = nullptr;
            }
            return rc;
        }
    }
    return PDDC_OK;
}

int pddc_comm_destroy(pddc_comm *c)
{
    if (!c)
        return PDDC_OK;
    hipSetDevice(c->device);
    if (c->side)
        hipStreamSynchronize(c->side);
    if (c->comm)
        ncclCommDestroy(c->comm);
    if (c->ev_ready)
        hipEventDestroy(c->ev_ready);
    for (int k = 0; k < 2; ++k)
        if (c->ev_done[k])
            hipEventDestroy(c->ev_done[k]);
    if (c->side)
        hipStreamDestroy(c->side);
    if (c->d_scratch)
        hipFree(c->d_scratch);
    delete c;
    return PDDC_OK;
}

int pddc_comm_rank(const pddc_comm *c) { return c ? c->rank : pddc_set_error_(PDDC_EINVAL, "null communicator"); }
int pddc_comm_size(const pddc_comm *c) { return c ? c->nranks : pddc_set_error_(PDDC_EINVAL, "null communicator"); }
int pddc_comm_device(const pddc_comm *c) { return c ? c->device : pddc_set_error_(PDDC_EINVAL, "null communicator"); }

int pddc_comm_group_start(void)
{
    NCCL_TRY(ncclGroupStart());
    ++g_group_depth;
    return PDDC_OK;
}

int pddc_comm_group_end(void)
{
    if (g_group_depth > 0)
        --g_group_depth;
    /* the list is taken BEFORE the call that may fail: a failing ncclGroupEnd must not leave communicators on it that a
     * later group would touch, possibly after they have been destroyed */
    std::vector<pddc_comm *> done;
    if (g_group_depth == 0)
        done.swap(g_group_pending);
    NCCL_TRY(ncclGroupEnd());
    if (g_group_depth == 0) {
        /* the grouped transfers are on their side streams now: mark their completion points */
        for (pddc_comm *c : done) {
            HIP_TRYM(hipSetDevice(c->device));
            HIP_TRYM(hipEventRecord(c->ev_done[(c->n_gathers - 1) & 1], c->side));
        }
    }
    return PDDC_OK;
}

int pddc_comm_bcast(pddc_comm *c, void *d_buf, size_t nbytes, int root, void *stream)
{
    if (!c || root < 0 || root >= c->nranks)
        return pddc_set_error_(PDDC_EINVAL, "bad broadcast arguments");
    if (nbytes == 0)
        return PDDC_OK;
    if (!d_buf)
        return pddc_set_error_(PDDC_EINVAL, "null device pointer");
    HIP_TRYM(hipSetDevice(c->device));
    NCCL_TRY(ncclBroadcast(d_buf, d_buf, nbytes, ncclUint8, root, c->comm, (hipStream_t)stream));
    return PDDC_OK;
}

int pddc_comm_bcast_host(pddc_comm *c, void *h_buf, size_t nbytes, int root)
{
    if (!c || root < 0 || root >= c->nranks)
        return pddc_set_error_(PDDC_EINVAL, "bad broadcast arguments");
    if (nbytes == 0)
        return PDDC_OK;
    if (!h_buf)
        return pddc_set_error_(PDDC_EINVAL, "null host pointer");
    HIP_TRYM(hipSetDevice(c->device));
    int rc = ensure_scratch(c, nbytes);
    if (rc)
        return rc;
    if (c->rank == root)
        HIP_TRYM(hipMemcpyAsync(c->d_scratch, h_buf, nbytes, hipMemcpyHostToDevice, c->side));
    NCCL_TRY(ncclBroadcast(c->d_scratch, c->d_scratch, nbytes, ncclUint8, root, c->comm, c->side));
    if (c->rank != root)
        HIP_TRYM(hipMemcpyAsync(h_buf, c->d_scratch, nbytes, hipMemcpyDeviceToHost, c->side));
    HIP_TRYM(hipStreamSynchronize(c->side));
    return PDDC_OK;
}

int pddc_comm_allreduce_max_f64(pddc_comm *c, double *h_val)
{
    if (!c || !h_val)
        return pddc_set_error_(PDDC_EINVAL, "bad allreduce arguments");
    HIP_TRYM(hipSetDevice(c->device));
    int rc = ensure_scratch(c, sizeof(double));
    if (rc)
        return rc;
    HIP_TRYM(hipMemcpyAsync(c->d_scratch, h_val, sizeof(double), hipMemcpyHostToDevice, c->side));
    NCCL_TRY(ncclAllReduce(c->d_scratch, c->d_scratch, 1, ncclFloat64, ncclMax, c->comm, c->side));
    HIP_TRYM(hipMemcpyAsync(h_val, c->d_scratch, sizeof(double), hipMemcpyDeviceToHost, c->side));
    HIP_TRYM(hipStreamSynchronize(c->side));
    return PDDC_OK;
}

int pddc_comm_barrier(pddc_comm *c)
{
    double v = 0.0;
    return pddc_comm_allreduce_max_f64(c, &v);
}

/* peer -> root.  The root posts one receive per peer and every peer one send, all inside
 * one group: the transfers run concurrently, one xGMI link each.  The root's own block
 * is a device-to-device copy on the same stream.                                       */
static int gather_on(pddc_comm *c, const void *d_send, size_t nbytes, void *d_recv, int root, hipStream_t s)
{
    if (c->rank == root) {
        uint8_t *dst = static_cast<uint8_t *>(d_recv);
        NCCL_TRY(ncclGroupStart());
        for (int r = 0; r < c->nranks; ++r) {
            if (r == root)
                continue;
            ncclResult_t e = ncclRecv(dst + (size_t)r * nbytes, nbytes, ncclUint8, r, c->comm, s);
            if (e != ncclSuccess) {
                ncclGroupEnd();
                return pddc_set_error_(PDDC_ECOMM, "ncclRecv from rank %d: %s", r, ncclGetErrorString(e));
            }
        }
        NCCL_TRY(ncclGroupEnd());
        if (dst + (size_t)root * nbytes != d_send)
            HIP_TRYM(hipMemcpyAsync(dst + (size_t)root * nbytes, d_send, nbytes, hipMemcpyDeviceToDevice, s));
    } else {
        NCCL_TRY(ncclSend(d_send, nbytes, ncclUint8, root, c->comm, s));
    }
    return PDDC_OK;
}

static int check_gather_args(pddc_comm *c, const void *d_send, size_t nbytes, void *d_recv, int root)
{
    if (!c || root < 0 || root >= c->nranks)
        return pddc_set_error_(PDDC_EINVAL, "bad gather arguments");
    if (nbytes && (!d_send || (c->rank == root && !d_recv)))
        return pddc_set_error_(PDDC_EINVAL, "null device pointer");
    return PDDC_OK;
}

int pddc_comm_gather(pddc_comm *c, const void *d_send, size_t nbytes, void *d_recv, int root, void *stream)
{
    int rc = check_gather_args(c, d_send, nbytes, d_recv, root);
    if (rc || nbytes == 0)
        return rc;
    HIP_TRYM(hipSetDevice(c->device));
    return gather_on(c, d_send, nbytes, d_recv, root, (hipStream_t)stream);
}

int pddc_comm_gather_async(pddc_comm *c, const void *d_send, size_t nbytes, void *d_recv, int root,
                           void *after_stream)
{
    int rc = check_gather_args(c, d_send, nbytes, d_recv, root);
    if (rc)
        return rc;
    HIP_TRYM(hipSetDevice(c->device));
    /* the side stream starts when everything queued on `after_stream` so far (the kernels
     * that produce d_send) has finished */
    HIP_TRYM(hipEventRecord(c->ev_ready, (hipStream_t)after_stream));
    HIP_TRYM(hipStreamWaitEvent(c->side, c->ev_ready, 0));
    if (nbytes && (rc = gather_on(c, d_send, nbytes, d_recv, root, c->side)))
        return rc;
    c->n_gathers++;
    if (g_group_depth > 0)
        g_group_pending.push_back(c);          /* recorded by pddc_comm_group_end, once RCCL has enqueued the group */
    else
        HIP_TRYM(hipEventRecord(c->ev_done[(c->n_gathers - 1) & 1], c->side));
    return PDDC_OK;
}

int pddc_comm_gather_fence(pddc_comm *c, void *stream)
{
    if (!c)
        return pddc_set_error_(PDDC_EINVAL, "null communicator");
    /* double buffering: the caller alternates two send buffers, so before it writes one again the
     * transfer BEFORE the most recent one must be done -- the most recent may still be running */
    if (c->n_gathers < 2)
        return PDDC_OK;
    HIP_TRYM(hipSetDevice(c->device));
    HIP_TRYM(hipStreamWaitEvent((hipStream_t)stream, c->ev_done[(c->n_gathers - 2) & 1], 0));
    return PDDC_OK;
}

int pddc_comm_gather_wait(pddc_comm *c)
{
    if (!c)
        return pddc_set_error_(PDDC_EINVAL, "null communicator");
    if (c->n_gathers == 0)
        return PDDC_OK;
    HIP_TRYM(hipSetDevice(c->device));
    HIP_TRYM(hipEventSynchronize(c->ev_done[(c->n_gathers - 1) & 1]));      /* the latest; the side stream is in order */
    return PDDC_OK;
}

/* ---- plan (de)serialisation: what the configuration broadcast carries -------- */
/* layout (little endian, 4-byte words):
 *   magic 'PDC1' | freg | flags | nstages | { decim, interp, ntaps } x nstages | taps ... */
static const uint32_t kPlanMagic = 0x31434450u;

size_t pddc_plan_pack(const pddc_stage_desc *stages, int nstages, uint32_t freg, uint32_t flags, void *buf,
                      size_t capacity)
{
    if (!stages || nstages < 1 || nstages > PDDC_MAX_STAGES) {
        pddc_set_error_(PDDC_EINVAL, "nstages must be 1..%d", PDDC_MAX_STAGES);
        return 0;
    }
    size_t need = 4 * (4 + 3 * (size_t)nstages);
    for (int i = 0; i < nstages; ++i) {
        if (stages[i].ntaps < 1 || stages[i].ntaps > PDDC_MAX_TAPS || !stages[i].taps) {
            pddc_set_error_(PDDC_EINVAL, "stage %d: bad taps", i);
            return 0;
        }
        need += 4 * (size_t)stages[i].ntaps;
    }
    if (!buf)
        return need;                         /* size query */
    if (capacity < need) {
        pddc_set_error_(PDDC_ECAPACITY, "plan needs %zu bytes, buffer has %zu", need, capacity);
        return 0;
    }
    uint32_t *w = static_cast<uint32_t *>(buf);
    *w++ = kPlanMagic;
    *w++ = freg;
    *w++ = flags;
    *w++ = (uint32_t)nstages;
    for (int i = 0; i < nstages; ++i) {
        *w++ = (uint32_t)stages[i].decim;
        *w++ = (uint32_t)(stages[i].interp > 1 ? stages[i].interp : 1);
        *w++ = (uint32_t)stages[i].ntaps;
    }
    for (int i = 0; i < nstages; ++i) {
        memcpy(w, stages[i].taps, 4 * (size_t)stages[i].ntaps);
        w += stages[i].ntaps;
    }
    return need;
}

int pddc_plan_unpack(const void *buf, size_t nbytes, pddc_stage_desc *stages, int *nstages, uint32_t *freg,
                     uint32_t *flags)
{
    if (!buf || !stages || !nstages || nbytes < 16)
        return pddc_set_error_(PDDC_EINVAL, "bad plan buffer");
    const uint32_t *w = static_cast<const uint32_t *>(buf);
    if (w[0] != kPlanMagic)
        return pddc_set_error_(PDDC_EINVAL, "not a plan buffer (magic %08x)", w[0]);
    const int n = (int)w[3];
    if (n < 1 || n > PDDC_MAX_STAGES || nbytes < 4 * (4 + 3 * (size_t)n))
        return pddc_set_error_(PDDC_EINVAL, "bad stage count %d", n);
    size_t off = 4 + 3 * (size_t)n;              /* words */
    for (int i = 0; i < n; ++i) {
        stages[i].decim = (int)w[4 + 3 * i];
        stages[i].interp = (int)w[5 + 3 * i];
        stages[i].ntaps = (int)w[6 + 3 * i];
        if (stages[i].ntaps < 1 || stages[i].ntaps > PDDC_MAX_TAPS || 4 * (off + (size_t)stages[i].ntaps) > nbytes)
            return pddc_set_error_(PDDC_EINVAL, "stage %d: tap count %d does not fit the buffer", i, stages[i].ntaps);
        stages[i].taps = reinterpret_cast<const float *>(w + off);     /* points INTO buf */
        off += (size_t)stages[i].ntaps;
    }
    *nstages = n;
    if (freg)
        *freg = w[1];
    if (flags)
        *flags = w[2];
    return PDDC_OK;
}

int pddc_comm_bcast_pipeline(pddc_comm *c, int root, const pddc_stage_desc *stages, int nstages, uint32_t freg,
                             uint32_t flags, pddc_pipeline **out)
{
    if (!c || !out || root < 0 || root >= c->nranks)
        return pddc_set_error_(PDDC_EINVAL, "bad arguments");
    *out = nullptr;
    /* fixed-size header first (the peers do not know the plan's size), then the plan */
    uint64_t size = 0;
    std::vector<uint8_t> buf;
    if (c->rank == root) {
        size = pddc_plan_pack(stages, nstages, freg, flags, nullptr, 0);
        if (size == 0)
            size = ~0ull;                        /* tell the peers the root failed */
        else {
            buf.resize(size);
            pddc_plan_pack(stages, nstages, freg, flags, buf.data(), buf.size());
        }
    }
    int rc = pddc_comm_bcast_host(c, &size, sizeof(size), root);
    if (rc)
        return rc;
    if (size == ~0ull || size > (1u << 20))
        return pddc_set_error_(PDDC_EINVAL, "root has no valid plan to broadcast");
    buf.resize(size);
    if ((rc = pddc_comm_bcast_host(c, buf.data(), buf.size(), root)))
        return rc;
    pddc_stage_desc sd[PDDC_MAX_STAGES];
    int n = 0;
    uint32_t fr = 0, fl = 0;
    if ((rc = pddc_plan_unpack(buf.data(), buf.size(), sd, &n, &fr, &fl)))
        return rc;
    if ((rc = pddc_pipeline_create(out, c->device, sd, n, fl)))
        return rc;
    return pddc_pipeline_set_freg(*out, fr);
}

} /* extern "C" */
