// comm.hip — native RCCL transport behind the C ABI (include/fluidsim.h, fs_comm_* / fs_slab_exchange).
//
// NOT in the reference (single wgpu device, src/renderer.rs:108-133).  A host in any language drives one
// process per GPU: rank 0 calls fs_comm_unique_id, ships the 128 bytes to the other ranks by whatever
// rendezvous it has, every rank calls fs_comm_init, and each step is
//     fs_slab_pack -> fs_slab_exchange -> fs_slab_step
// with the exchange issued as ONE grouped ncclSend/ncclRecv set on the simulation's own HIP stream (so it is
// ordered after the pack kernels and before the unpack kernel without any host synchronisation).
// xGMI is point to point: a slab only ever talks to its two neighbours, one fixed-size message each way.
//
// librccl is bound lazily with dlopen (FS_RCCL_LIB, then librccl.so.1, then /opt/rocm/lib/librccl.so.1): a process
// that already carries an RCCL (e.g. PyTorch's) keeps using that one, and single-GPU users never load it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/fluidsim.h"

namespace fsd { void set_last_error(const std::string& msg); }

struct fs_sim;
extern "C" void* fs_stream(const fs_sim* sim);
extern "C" size_t fs_slab_message_bytes(const fs_sim* sim);

namespace {

typedef struct { char internal[128]; } nccl_unique_id;       // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES 128)
typedef void* nccl_comm;
enum { NCCL_SUCCESS = 0, NCCL_INT8 = 0, NCCL_UINT8 = 1, NCCL_UINT32 = 3, NCCL_UINT64 = 5, NCCL_FLOAT32 = 7 };
enum { NCCL_SUM = 0, NCCL_MAX = 2 };

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(nccl_unique_id*) = nullptr;
    int (*CommInitRank)(nccl_comm*, int, nccl_unique_id, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, nccl_comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};

Rccl* rccl() {
    static Rccl R;
    static bool tried = false;
    if (tried) return R.lib ? &R : nullptr;
    tried = true;
    // FS_RCCL_LIB, when set, is the ONLY candidate (so a deployment can pin its RCCL, and tests can make it fail)
    const char* pinned = getenv("FS_RCCL_LIB");
    const char* names[] = {pinned, pinned ? nullptr : "librccl.so.1", pinned ? nullptr : "librccl.so",
                           pinned ? nullptr : "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n) continue;
        R.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (R.lib) break;
        R.why = dlerror();
    }
    if (!R.lib) return nullptr;
#define BIND(field, sym)                                                          \
    *(void**)(&R.field) = dlsym(R.lib, sym);                                      \
    if (!R.field) { R.why = std::string("missing symbol ") + sym; R.lib = nullptr; return nullptr; }
    BIND(GetUniqueId, "ncclGetUniqueId")
    BIND(CommInitRank, "ncclCommInitRank")
    BIND(CommDestroy, "ncclCommDestroy")
    BIND(GroupStart, "ncclGroupStart")
    BIND(GroupEnd, "ncclGroupEnd")
    BIND(Send, "ncclSend")
    BIND(Recv, "ncclRecv")
    BIND(AllReduce, "ncclAllReduce")
    BIND(GetErrorString, "ncclGetErrorString")
#undef BIND
    return &R;
}

fs_status comm_fail(const std::string& msg) {
    fsd::set_last_error(msg);
    return FS_ERR_COMM;
}

#define FS_NCCL(R, expr)                                                                              \
    do {                                                                                              \
        const int rc__ = (expr);                                                                      \
        if (rc__ != NCCL_SUCCESS) return comm_fail(std::string(#expr) + ": " + (R)->GetErrorString(rc__)); \
    } while (0)

}  // namespace

extern "C" {

fs_status fs_comm_unique_id(uint8_t id[FS_COMM_ID_BYTES]) {
    if (!id) { fsd::set_last_error("null argument"); return FS_ERR_INVALID; }
    Rccl* R = rccl();
    if (!R) return comm_fail("librccl not available (dlopen failed; FS_RCCL_LIB pins the library)");
    nccl_unique_id u;
    FS_NCCL(R, R->GetUniqueId(&u));
    std::memcpy(id, u.internal, sizeof u.internal);
    return FS_OK;
}

fs_status fs_comm_init(int device, int rank, int world, const uint8_t id[FS_COMM_ID_BYTES], fs_comm** out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) { fsd::set_last_error("bad argument"); return FS_ERR_INVALID; }
    *out = nullptr;
    Rccl* R = rccl();
    if (!R) return comm_fail("librccl not available");
    if (hipSetDevice(device) != hipSuccess) { fsd::set_last_error("hipSetDevice failed"); return FS_ERR_DEVICE; }
    nccl_unique_id u;
    std::memcpy(u.internal, id, sizeof u.internal);
    nccl_comm c = nullptr;
    FS_NCCL(R, R->CommInitRank(&c, world, u, rank));
    *out = (fs_comm*)c;
    return FS_OK;
}

void fs_comm_destroy(fs_comm* comm) {
    Rccl* R = rccl();
    if (R && comm) (void)R->CommDestroy((nccl_comm)comm);
}

fs_status fs_slab_exchange(fs_sim* sim, fs_comm* comm, int left_rank, int right_rank, const void* send_left,
                           const void* send_right, void* recv_left, void* recv_right) {
    if (!sim || !comm) { fsd::set_last_error("null argument"); return FS_ERR_INVALID; }
    const size_t bytes = fs_slab_message_bytes(sim);
    if (bytes == 0) { fsd::set_last_error("not a slab handle"); return FS_ERR_INVALID; }
    if ((left_rank >= 0 && (!send_left || !recv_left)) || (right_rank >= 0 && (!send_right || !recv_right))) {
        fsd::set_last_error("missing message buffer for a present neighbour");
        return FS_ERR_INVALID;
    }
    if (left_rank < 0 && right_rank < 0) return FS_OK;          // a single slab: nothing to exchange
    Rccl* R = rccl();
    if (!R) return comm_fail("librccl not available");
    // overlapped step: on the handle's exchange stream, behind the pack (fs_slab_comm_begin) and ahead of the boundary
    // strips (fs_slab_comm_end) — the interior columns' kernels run beside it on the simulation's stream; serial step: on the
    // simulation's stream itself
    const bool overlapped = fs_slab_overlapped(sim) != 0;
    hipStream_t st = (hipStream_t)(overlapped ? fs_slab_comm_stream(sim) : fs_stream(sim));
    if (overlapped) { const fs_status r = fs_slab_comm_begin(sim); if (r != FS_OK) return r; }
    nccl_comm c = (nccl_comm)comm;
    // one group: both directions progress together (no send/recv ordering deadlock between neighbours)
    FS_NCCL(R, R->GroupStart());
    int rc = NCCL_SUCCESS;
    if (right_rank >= 0) {
        if (rc == NCCL_SUCCESS) rc = R->Send(send_right, bytes, NCCL_UINT8, right_rank, c, st);
        if (rc == NCCL_SUCCESS) rc = R->Recv(recv_right, bytes, NCCL_UINT8, right_rank, c, st);
    }
    if (left_rank >= 0) {
        if (rc == NCCL_SUCCESS) rc = R->Send(send_left, bytes, NCCL_UINT8, left_rank, c, st);
        if (rc == NCCL_SUCCESS) rc = R->Recv(recv_left, bytes, NCCL_UINT8, left_rank, c, st);
    }
    const int rc_end = R->GroupEnd();
    if (rc != NCCL_SUCCESS) return comm_fail(std::string("ncclSend/ncclRecv: ") + R->GetErrorString(rc));
    if (rc_end != NCCL_SUCCESS) return comm_fail(std::string("ncclGroupEnd: ") + R->GetErrorString(rc_end));
    if (overlapped) return fs_slab_comm_end(sim);
    return FS_OK;
}

fs_status fs_comm_allreduce(fs_sim* sim, fs_comm* comm, void* device_buf, size_t count, int dtype, int op) {
    if (!sim || !comm || (!device_buf && count)) { fsd::set_last_error("null argument"); return FS_ERR_INVALID; }
    int dt, o;
    switch (dtype) { case FS_COMM_U32: dt = NCCL_UINT32; break; case FS_COMM_U64: dt = NCCL_UINT64; break;
                     case FS_COMM_F32: dt = NCCL_FLOAT32; break; default: fsd::set_last_error("unknown dtype"); return FS_ERR_INVALID; }
    switch (op) { case FS_COMM_SUM: o = NCCL_SUM; break; case FS_COMM_MAX: o = NCCL_MAX; break;
                  default: fsd::set_last_error("unknown op"); return FS_ERR_INVALID; }
    Rccl* R = rccl();
    if (!R) return comm_fail("librccl not available");
    FS_NCCL(R, R->AllReduce(device_buf, device_buf, count, dt, o, (nccl_comm)comm, (hipStream_t)fs_stream(sim)));
    return FS_OK;
}

}  // extern "C"
