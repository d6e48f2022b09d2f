// trk_exchange.hip -- the batch-sharded planner's ONE exchange step as a peer-to-peer mailbox ("alternative =
// peer-to-peer write of 8 partials + local sum"), next to the RCCL all-reduce of torch_robotics_amd/distributed.py.
//
// What is exchanged is tiny (2 kB for Panda at horizon 64, 7.7 kB for config 5) and an evaluation is ~10 us, so the cost of an
// exchange is latency and launch overhead, not bytes: a ring all-reduce is 2 (N - 1) dependent xGMI hops plus the collective's own
// kernel and host enqueue (~50 - 80 us end to end through RCCL at this size).  Here every rank STORES its packed row straight into a
// slot of every peer's mailbox (device memory mapped through hipIpc*; one xGMI hop, all peers in parallel), then a sequence flag;
// the consumer waits for the N flags of its own mailbox and adds the N rows IN RANK ORDER -- every rank computes the same bits, and
// the same bits on every run.  Two single-workgroup kernels per exchange -- SEND (stores + flag, never waits) and RECEIVE (waits for
// the flags, adds the rows) -- so that a planner can send right after the evaluation whose sums travel and receive several
// evaluations later, when the peers' rows have long arrived: the wait is then off the critical path without a second stream.  Both
// take their sequence numbers from device counters, so they can be captured into a hipGraph and replayed.
//
// Memory: the mailbox is allocated uncached / fine-grained (hipExtMallocWithFlags) so that a peer's stores over xGMI are visible to
// the owner's loads without a kernel boundary; all mailbox accesses are system-scope atomics (sc0 sc1 loads / stores) on top of that.
// Slot reuse: a rank alternates send k, receive k in stream order.  When it sends k it has finished receive k - 1, which needed every
// peer's send k - 1, which followed that peer's receive k - 2: with >= 2 slots nobody overwrites a row that has not been read.
// LOOK-AHEAD RULE (enforced on the host, trk_mailbox_send): with `o` sends not yet received on this rank, the next send m has seen
// receive m - 1 - o, hence every peer's send m - 1 - o, which that peer issued after ITS receive m - 2 - 2 o; the send overwrites the
// row of sequence m - n_slots in every mailbox, which every peer must have read: m - n_slots <= m - 2 - 2 o, i.e.
// o <= (n_slots - 2) / 2.  Two slots: strict alternation; four slots: one send of look-ahead (send k + 1 before receive k -- the
// bench's step graphs and tools/mailbox_soak.py); six: two.  A violating send is refused with TRK_ERR_INVALID_ARG instead of
// overwriting a row a peer has not read.
// TIME-OUT: a receive whose flags do not arrive within TRK_MAILBOX_TIMEOUT_S writes NaN to EVERY element of its output (never a
// plausible partial sum with a late peer's stale rows) and counts the event in counter[1], which is sticky (trk_mailbox_status).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include "../../include/trk.h"
#include "trk_launch.h"

#define TRK_MAILBOX_MAX_RANKS 16
#define TRK_MAILBOX_FLAG_STRIDE 16            // uint32 words between two flags: one 64-byte line per (slot, writer)
#define TRK_MAILBOX_MAX_FLOATS (1 << 22)      // 16 MiB per row: far beyond any packed exchange (7.7 kB for config 5), keeps every offset sane

struct TrkMailbox {
    int world = 0, rank = 0, n_floats = 0, n_slots = 0, stride = 0;
    size_t flag_off = 0;                      // in 4-byte words from the base
    size_t bytes = 0;
    int alloc_kind = -1;                      // 0 uncached, 1 fine-grained, 2 plain hipMalloc
    void* local = nullptr;
    void* peers[TRK_MAILBOX_MAX_RANKS] = {};
    unsigned* counter = nullptr;              // [0] rows sent, [1] time-outs seen, [2] sums received
    bool connected = false;
    double timeout_s = 5.0;
    long long host_sends = 0, host_recvs = 0; // launches issued from the host (a captured launch counts once: a graph must be balanced)
};

namespace {
struct MailboxArgs {
    unsigned* base[TRK_MAILBOX_MAX_RANKS];    // every rank's mailbox as mapped HERE (base[rank] = the local one)
    int world, rank, n, stride, n_slots;
    size_t flag_off;                           // in 4-byte words (n_slots x world x stride: may exceed 32 bits for long rows)
    unsigned* counter;
    const float* packed;
    float* out;
    unsigned long long timeout_ticks;          // of the 100 MHz wall clock
};

__device__ __forceinline__ void st_sys(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ unsigned ld_sys(const unsigned* p) { return __hip_atomic_load(const_cast<unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// counter[0]: rows sent, counter[1]: time-outs seen, counter[2]: sums received.  A rank alternates send k, receive k in stream order.
__global__ void __launch_bounds__(256)
k_mailbox_send(MailboxArgs a) {
    __shared__ unsigned s_seq;
    const int tid = threadIdx.x;
    if (tid == 0) { const unsigned v = a.counter[0] + 1u; a.counter[0] = v; s_seq = v; }
    __syncthreads();
    const unsigned seq = s_seq;
    const unsigned slot = seq % (unsigned)a.n_slots;
    const size_t row = ((size_t)slot * a.world + a.rank) * a.stride;
    // my row into slot [slot][rank] of every mailbox (my own included), then the flag
    for (int p = 0; p < a.world; ++p) {
        unsigned* dst = a.base[p] + row;
        for (int i = tid; i < a.n; i += 256) st_sys(dst + i, __float_as_uint(a.packed[i]));
    }
    __threadfence_system();
    __syncthreads();
    if (tid < a.world)
        __hip_atomic_store(a.base[tid] + a.flag_off + ((size_t)slot * a.world + a.rank) * TRK_MAILBOX_FLAG_STRIDE, seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(256)
k_mailbox_recv(MailboxArgs a) {
    __shared__ unsigned s_seq;
    __shared__ int s_timed_out;
    const int tid = threadIdx.x;
    if (tid == 0) { const unsigned v = a.counter[2] + 1u; a.counter[2] = v; s_seq = v; s_timed_out = 0; }
    __syncthreads();
    const unsigned seq = s_seq;
    const unsigned slot = seq % (unsigned)a.n_slots;
    // the flags of all writers in MY mailbox
    if (tid < a.world) {
        unsigned* f = a.base[a.rank] + a.flag_off + ((size_t)slot * a.world + tid) * TRK_MAILBOX_FLAG_STRIDE;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            if (wall_clock64() - t0 > a.timeout_ticks) { s_timed_out = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    __threadfence_system();
    if (s_timed_out) {                         // a dead or slow peer: NaN everywhere -- never a plausible sum with stale rows in it
        if (tid == 0) a.counter[1] += 1u;
        for (int i = tid; i < a.n; i += 256) a.out[i] = __uint_as_float(0x7fc00000u);
        return;
    }
    // sum in rank order: the same association on every rank
    const unsigned* mine = a.base[a.rank] + (size_t)slot * a.world * a.stride;
    for (int i = tid; i < a.n; i += 256) {
        float acc = __uint_as_float(ld_sys(mine + i));
        for (int r = 1; r < a.world; ++r) acc += __uint_as_float(ld_sys(mine + (size_t)r * a.stride + i));
        a.out[i] = acc;
    }
}

int alloc_mailbox(TrkMailbox* mb) {
    const char* forced = std::getenv("TRK_MAILBOX_ALLOC");
    const unsigned kinds[3] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained, 0u};
    const char* names[3] = {"uncached", "finegrained", "plain"};
    hipError_t last = hipSuccess;
    for (int k = 0; k < 3; ++k) {
        if (forced && std::strcmp(forced, names[k]) != 0) continue;
        void* p = nullptr;
        hipError_t e = kinds[k] ? hipExtMallocWithFlags(&p, mb->bytes, kinds[k]) : hipMalloc(&p, mb->bytes);
        if (e == hipSuccess) {
            hipIpcMemHandle_t h;
            e = mb->world > 1 ? hipIpcGetMemHandle(&h, p) : hipSuccess;        // an allocation that cannot be shared is of no use
            if (e == hipSuccess) e = hipMemset(p, 0, mb->bytes);
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e == hipSuccess) { mb->local = p; mb->alloc_kind = k; return TRK_OK; }
            (void)hipFree(p);
        }
        (void)hipGetLastError();
        last = e;
    }
    return trk_hip_fail((int)last, "trk_mailbox_create: no shareable device allocation (uncached / fine-grained / plain all failed)");
}
}  // namespace

extern "C" {

int trk_mailbox_create(int32_t world, int32_t rank, int32_t n_floats, int32_t n_slots, TrkMailbox** out) {
    if (!out || world < 1 || world > TRK_MAILBOX_MAX_RANKS || rank < 0 || rank >= world || n_floats < 1 || n_floats > TRK_MAILBOX_MAX_FLOATS ||
        n_slots < 2 || n_slots > 64)
        return trk_fail(TRK_ERR_INVALID_ARG, "trk_mailbox_create: bad argument (1 <= world <= 16, 0 <= rank < world, 1 <= n_floats <= 2^22, 2 <= n_slots <= 64)");
    int rc = trk_ensure_init();
    if (rc) return rc;
    TrkMailbox* mb = new (std::nothrow) TrkMailbox();
    if (!mb) return trk_fail(TRK_ERR_HIP, "trk_mailbox_create: out of host memory");
    mb->world = world; mb->rank = rank; mb->n_floats = n_floats; mb->n_slots = n_slots;
    mb->stride = (n_floats + 15) / 16 * 16;                                  // rows start on 64-byte lines
    mb->flag_off = (size_t)n_slots * world * mb->stride;
    mb->bytes = 4 * (mb->flag_off + (size_t)n_slots * world * TRK_MAILBOX_FLAG_STRIDE);
    if (const char* t = std::getenv("TRK_MAILBOX_TIMEOUT_S")) { const double v = std::atof(t); if (v > 0.0) mb->timeout_s = v; }
    rc = alloc_mailbox(mb);
    if (rc) { delete mb; return rc; }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&mb->counter), 64);
    if (e == hipSuccess) e = hipMemset(mb->counter, 0, 64);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(mb->local); delete mb; return trk_hip_fail((int)e, "trk_mailbox_create: counter"); }
    mb->peers[rank] = mb->local;
    mb->connected = world == 1;
    *out = mb;
    return TRK_OK;
}

int trk_mailbox_ipc_handle(const TrkMailbox* mb, void* handle64) {
    if (!mb || !handle64) return trk_fail(TRK_ERR_INVALID_ARG, "trk_mailbox_ipc_handle: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == TRK_MAILBOX_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t h;
    hipError_t e = hipIpcGetMemHandle(&h, mb->local);
    if (e != hipSuccess) return trk_hip_fail((int)e, "trk_mailbox_ipc_handle: hipIpcGetMemHandle");
    std::memcpy(handle64, &h, sizeof(h));
    return TRK_OK;
}

int trk_mailbox_connect(TrkMailbox* mb, const void* handles) {
    if (!mb || !handles) return trk_fail(TRK_ERR_INVALID_ARG, "trk_mailbox_connect: null argument");
    if (mb->connected) return TRK_OK;
    for (int p = 0; p < mb->world; ++p) {
        if (p == mb->rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, static_cast<const char*>(handles) + (size_t)p * sizeof(h), sizeof(h));
        void* ptr = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int k = 0; k < p; ++k) if (k != mb->rank && mb->peers[k]) { (void)hipIpcCloseMemHandle(mb->peers[k]); mb->peers[k] = nullptr; }
            return trk_hip_fail((int)e, "trk_mailbox_connect: hipIpcOpenMemHandle");
        }
        mb->peers[p] = ptr;
    }
    mb->connected = true;
    return TRK_OK;
}

namespace {
MailboxArgs mailbox_args(const TrkMailbox* mb, const float* packed, float* out) {
    MailboxArgs a;
    for (int p = 0; p < TRK_MAILBOX_MAX_RANKS; ++p) a.base[p] = static_cast<unsigned*>(p < mb->world ? mb->peers[p] : nullptr);
    a.world = mb->world; a.rank = mb->rank; a.n = mb->n_floats; a.stride = mb->stride; a.n_slots = mb->n_slots;
    a.flag_off = mb->flag_off; a.counter = mb->counter; a.packed = packed; a.out = out;
    a.timeout_ticks = (unsigned long long)(mb->timeout_s * 1e8);
    return a;
}
}  // namespace

int trk_mailbox_send(TrkMailbox* mb, const float* packed, trk_stream_t stream) {
    if (!mb || !packed) return trk_fail(TRK_ERR_INVALID_ARG, "trk_mailbox_send: null argument");
    if (!mb->connected) return trk_fail(TRK_ERR_INVALID_ARG, "trk_mailbox_send: trk_mailbox_connect has not been called");
    if (2 * (mb->host_sends - mb->host_recvs) > mb->n_slots - 2)
        return trk_fail(TRK_ERR_INVALID_ARG, "trk_mailbox_send: too many sends ahead of their receives for this mailbox's slots (send k + a may precede "
                                             "receive k only for a <= (n_slots - 2) / 2): the send could overwrite a row a peer has not read");
    mb->host_sends += 1;
    hipLaunchKernelGGL(k_mailbox_send, dim3(1), dim3(256), 0, (hipStream_t)stream, mailbox_args(mb, packed, nullptr));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return trk_hip_fail((int)e, "trk_mailbox_send: launch");
    return TRK_OK;
}

int trk_mailbox_recv(TrkMailbox* mb, float* out, trk_stream_t stream) {
    if (!mb || !out) return trk_fail(TRK_ERR_INVALID_ARG, "trk_mailbox_recv: null argument");
    if (!mb->connected) return trk_fail(TRK_ERR_INVALID_ARG, "trk_mailbox_recv: trk_mailbox_connect has not been called");
    if (mb->host_recvs >= mb->host_sends) return trk_fail(TRK_ERR_INVALID_ARG, "trk_mailbox_recv: nothing has been sent that is not received yet");
    mb->host_recvs += 1;
    hipLaunchKernelGGL(k_mailbox_recv, dim3(1), dim3(256), 0, (hipStream_t)stream, mailbox_args(mb, nullptr, out));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return trk_hip_fail((int)e, "trk_mailbox_recv: launch");
    return TRK_OK;
}

int trk_mailbox_exchange(TrkMailbox* mb, const float* packed, float* out, trk_stream_t stream) {
    if (!mb || !packed || !out) return trk_fail(TRK_ERR_INVALID_ARG, "trk_mailbox_exchange: null argument");
    int rc = trk_mailbox_send(mb, packed, stream);
    return rc ? rc : trk_mailbox_recv(mb, out, stream);
}

int trk_mailbox_status(const TrkMailbox* mb, int64_t* n_exchanges, int64_t* n_timeouts, int32_t* alloc_kind) {
    if (!mb) return trk_fail(TRK_ERR_INVALID_ARG, "trk_mailbox_status: null mailbox");
    unsigned host[3] = {0, 0, 0};
    hipError_t e = hipMemcpy(host, mb->counter, sizeof(host), hipMemcpyDeviceToHost);     // synchronises with the device
    if (e != hipSuccess) return trk_hip_fail((int)e, "trk_mailbox_status: hipMemcpy");
    if (n_exchanges) *n_exchanges = host[0];
    if (n_timeouts) *n_timeouts = host[1];
    if (alloc_kind) *alloc_kind = mb->alloc_kind;
    return TRK_OK;
}

void trk_mailbox_destroy(TrkMailbox* mb) {
    if (!mb) return;
    for (int p = 0; p < mb->world; ++p)
        if (p != mb->rank && mb->peers[p]) (void)hipIpcCloseMemHandle(mb->peers[p]);
    if (mb->local) (void)hipFree(mb->local);
    if (mb->counter) (void)hipFree(mb->counter);
    (void)hipGetLastError();
    delete mb;
}

}  // extern "C"
