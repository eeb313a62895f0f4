// mnv_comm.cpp -- the one collective of the path: gather of the ranks' tile buffers to the root over RCCL / xGMI.
//
// rays are independent and the tree is read-only, so the frame is partitioned (mnv_partition) and the only
// exchange is the gather of the compact per-rank tile buffers to rank 0, which un-permutes them (mnv_assemble_tiles).  The
// reference has no multi-GPU code (its VolumeRenderer::Impl::render, src/renderer/cuda_renderer.cpp:68-163, drives one device);
// this is the north star's "RCCL gather of RGBA tiles over xGMI", one process per GPU.
//
// The gather is grouped ncclSend / ncclRecv -- the root posts world - 1 receives straight into its [world][bytes] table, every
// other rank one send -- because the pattern is root-inbound over seven point-to-point links, not a ring.
//
// RCCL is bound at the first mnv_comm_* call (dlopen of librccl.so.1; a copy that is already in the process, e.g. PyTorch's, is
// reused): single-GPU users of libmnv.so do not load the 570 MB library, and two RCCL copies never meet in one process.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/mnv.h"
#include "mnv_error.h"

namespace {

// the slice of rccl.h this file uses (RCCL 2.2x ABI: ncclUniqueId is 128 bytes, ncclComm_t an opaque pointer)
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[MNV_COMM_ID_BYTES];
} ncclUniqueId;
typedef int ncclResult_t;
enum { ncclSuccess = 0 };
enum { ncclUint8 = 1 };  // rccl.h: ncclInt8 = 0, ncclUint8 = 1

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string error;
};

Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
#ifdef MNV_TEST_HOOKS
        // Test hook, compiled only into the -DMNV_TEST_HOOKS build (testhooks/libmnv.so, what the rehearsal tests load): MNV_RCCL_LIBRARY
        // names a library with the same eight entry points (tests/shim/fake_rccl.cpp lets the world > 1 paths run with several ranks
        // on one GPU, which RCCL refuses).  The shipped libmnv.so binds librccl and nothing else.
        if (const char *over = getenv("MNV_RCCL_LIBRARY")) {
            r.handle = dlopen(over, RTLD_NOW | RTLD_LOCAL);
            if (!r.handle) {
                r.error = std::string("MNV_RCCL_LIBRARY: cannot load ") + over + ": " + (dlerror() ? dlerror() : "?");
                return;
            }
        }
#endif
        // a copy that is already loaded wins (PyTorch ships its own librccl.so with the same soname)
        for (const char *n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char *n : names)
            if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!r.handle) {
            r.error = std::string("cannot load librccl.so.1: ") + (dlerror() ? dlerror() : "not found");
            return;
        }
        auto sym = [&](const char *name) -> void * {
            void *p = dlsym(r.handle, name);
            if (!p && r.error.empty()) r.error = std::string("librccl.so.1 lacks ") + name;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
    });
    return r;
}

int need_rccl() {
    Rccl &r = rccl();
    if (!r.error.empty()) return mnv::set_error(MNV_E_NO_RCCL, r.error);
    return MNV_OK;
}

int check_nccl(ncclResult_t e, const char *what) {
    if (e == ncclSuccess) return MNV_OK;
    const char *s = rccl().GetErrorString ? rccl().GetErrorString(e) : "?";
    return mnv::set_error(MNV_E_RCCL, std::string(what) + ": RCCL error " + std::to_string(e) + " (" + s + ")");
}

}  // namespace

struct mnv_comm {
    ncclComm_t comm = nullptr;
    int32_t rank = 0, world = 1, device = 0;
};

extern "C" {

int mnv_comm_get_unique_id(void *id_out) {
    if (!id_out) return mnv::set_error(MNV_E_INVALID, "id_out is null");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    std::memset(&id, 0, sizeof(id));
    if (int rc = check_nccl(rccl().GetUniqueId(&id), "ncclGetUniqueId")) return rc;
    std::memcpy(id_out, &id, sizeof(id));
    return MNV_OK;
}

int mnv_comm_init_rank(const void *id, int32_t world, int32_t rank, mnv_comm **out) {
    if (!id || !out || world < 1 || rank < 0 || rank >= world) return mnv::set_error(MNV_E_INVALID, "need an id, world >= 1 and 0 <= rank < world");
    if (int rc = need_rccl()) return rc;
    mnv_comm *c = new mnv_comm();
    c->rank = rank;
    c->world = world;
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        return mnv::set_error(MNV_E_NO_DEVICE, "no current HIP device");
    }
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    if (int rc = check_nccl(rccl().CommInitRank(&c->comm, world, uid, rank), "ncclCommInitRank")) {
        delete c;
        return rc;
    }
    *out = c;
    return MNV_OK;
}

int32_t mnv_comm_rank(const mnv_comm *c) { return c ? c->rank : -1; }
int32_t mnv_comm_world(const mnv_comm *c) { return c ? c->world : 0; }

int32_t mnv_comm_rccl_version(void) {
    if (need_rccl() != MNV_OK) return 0;
    int v = 0;
    if (!rccl().GetVersion || rccl().GetVersion(&v) != ncclSuccess) return 0;
    return v;
}

int mnv_gather_tiles(mnv_comm *c, const void *local, void *gathered, size_t bytes_per_rank, int32_t root, void *hip_stream) {
    if (!c || !local || root < 0 || root >= c->world) return mnv::set_error(MNV_E_INVALID, "invalid gather arguments");
    if (c->rank == root && !gathered) return mnv::set_error(MNV_E_INVALID, "the root needs the [world][bytes_per_rank] table");
    if (bytes_per_rank == 0) return MNV_OK;
    Rccl &r = rccl();
    hipStream_t stream = (hipStream_t)hip_stream;
    int rc = check_nccl(r.GroupStart(), "ncclGroupStart");
    if (rc) return rc;
    if (c->rank == root) {
        uint8_t *table = static_cast<uint8_t *>(gathered);
        for (int32_t p = 0; p < c->world && !rc; ++p) {
            if (p == root && c->world > 1) continue;  // own share: device copy below
            rc = check_nccl(r.Recv(table + (size_t)p * bytes_per_rank, bytes_per_rank, ncclUint8, p, c->comm, stream), "ncclRecv");
        }
        // world 1: the root's share goes through a send / receive to itself, so that a one-GPU box runs the RCCL path for real
        if (c->world == 1 && !rc) rc = check_nccl(r.Send(local, bytes_per_rank, ncclUint8, root, c->comm, stream), "ncclSend");
    } else {
        rc = check_nccl(r.Send(local, bytes_per_rank, ncclUint8, root, c->comm, stream), "ncclSend");
    }
    const int rc_end = check_nccl(r.GroupEnd(), "ncclGroupEnd");
    if (rc) return rc;
    if (rc_end) return rc_end;
    if (c->rank == root && c->world > 1) {
        const hipError_t e = hipMemcpyAsync(static_cast<uint8_t *>(gathered) + (size_t)root * bytes_per_rank, local, bytes_per_rank, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return mnv::set_error((int)e, std::string("gather: copy of the root's share: ") + hipGetErrorString(e));
    }
    return MNV_OK;
}

int mnv_allgather(mnv_comm *c, void *table, size_t bytes_per_rank, void *hip_stream) {
    if (!c || !table) return mnv::set_error(MNV_E_INVALID, "invalid all-gather arguments");
    if (bytes_per_rank == 0 || c->world == 1) return MNV_OK;
    Rccl &r = rccl();
    hipStream_t stream = (hipStream_t)hip_stream;
    uint8_t *t = static_cast<uint8_t *>(table);
    int rc = check_nccl(r.GroupStart(), "ncclGroupStart");
    if (rc) return rc;
    for (int32_t p = 0; p < c->world && !rc; ++p) {
        if (p == c->rank) continue;
        rc = check_nccl(r.Send(t + (size_t)c->rank * bytes_per_rank, bytes_per_rank, ncclUint8, p, c->comm, stream), "ncclSend");
        if (!rc) rc = check_nccl(r.Recv(t + (size_t)p * bytes_per_rank, bytes_per_rank, ncclUint8, p, c->comm, stream), "ncclRecv");
    }
    const int rc_end = check_nccl(r.GroupEnd(), "ncclGroupEnd");
    return rc ? rc : rc_end;
}

void mnv_comm_destroy(mnv_comm *c) {
    if (!c) return;
    if (c->comm && rccl().CommDestroy) (void)rccl().CommDestroy(c->comm);
    delete c;
}

}  // extern "C"
