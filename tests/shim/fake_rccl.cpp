// fake_rccl.cpp -- TEST INFRASTRUCTURE, not product: a stand-in for the eight librccl.so.1 entry points mnv_comm.cpp binds, so that the
// world > 1 code paths of libmnv.so and mnv_render (rendezvous, partition, mnv_gather_tiles with several peers, un-permute, file output)
// can run with several ranks on ONE GPU -- RCCL itself refuses two ranks on one device ("Duplicate GPU detected").  The transport is
// host staged through a POSIX shared-memory segment: ncclSend copies device -> segment and raises a flag, ncclRecv waits for the flag and
// copies segment -> device; both block the calling thread (fine for a test, useless for performance).  What this does NOT cover is RCCL's
// own transport over xGMI; that needs a multi-GPU machine.  Selected with MNV_RCCL_LIBRARY=<this .so> (csrc/mnv_comm.cpp).
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace {

constexpr int kMaxRanks = 16;
constexpr size_t kSlotBytes = (size_t)96 << 20;  // per (sender, receiver) pair in flight: enough for 64 RGBA8 1080p frames / 2 ranks

struct Mailbox {
    std::atomic<uint64_t> seq_written;   // messages the sender has placed
    std::atomic<uint64_t> seq_consumed;  // messages the receiver has taken
    std::atomic<uint64_t> bytes;
};

struct Segment {
    std::atomic<int> arrived;
    Mailbox box[kMaxRanks][kMaxRanks];   // [sender][receiver]
};

struct Comm {
    int rank, world;
    Segment *seg;
    uint8_t *data;  // [sender][receiver][kSlotBytes], mapped lazily by both sides
    size_t map_bytes;
    char name[64];
};

struct Op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    Comm *comm;
    hipStream_t stream;
};
thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;

uint8_t *slot(Comm *c, int sender, int receiver) { return c->data + ((size_t)sender * c->world + receiver) * kSlotBytes; }

int run(const Op &op) {
    Comm *c = op.comm;
    if (op.bytes > kSlotBytes) {
        fprintf(stderr, "fake_rccl: message of %zu bytes exceeds the %zu-byte slot\n", op.bytes, kSlotBytes);
        return 1;
    }
    if (op.send) {
        Mailbox &m = c->seg->box[c->rank][op.peer];
        while (m.seq_written.load() != m.seq_consumed.load()) usleep(50);  // one message in flight per pair
        if (hipStreamSynchronize(op.stream) != hipSuccess) return 1;
        if (hipMemcpy(slot(c, c->rank, op.peer), op.buf, op.bytes, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        m.bytes.store(op.bytes);
        m.seq_written.fetch_add(1);
    } else {
        Mailbox &m = c->seg->box[op.peer][c->rank];
        while (m.seq_written.load() == m.seq_consumed.load()) usleep(50);
        if (m.bytes.load() != op.bytes) {
            fprintf(stderr, "fake_rccl: rank %d expected %zu bytes from %d, got %llu\n", c->rank, op.bytes, op.peer, (unsigned long long)m.bytes.load());
            return 1;
        }
        if (hipStreamSynchronize(op.stream) != hipSuccess) return 1;
        if (hipMemcpy(op.buf, slot(c, op.peer, c->rank), op.bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
        m.seq_consumed.fetch_add(1);
    }
    return 0;
}

}  // namespace

extern "C" {

typedef struct {
    char internal[128];
} ncclUniqueId;

int ncclGetVersion(int *v) {
    *v = 29999;  // recognisably not a real RCCL
    return 0;
}
const char *ncclGetErrorString(int) { return "fake_rccl error"; }

int ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/mnv_fake_rccl_%d_%ld", (int)getpid(), (long)random());
    return 0;
}

int ncclCommInitRank(void **out, int world, ncclUniqueId id, int rank) {
    if (world < 1 || world > kMaxRanks) return 1;
    Comm *c = new Comm();
    c->rank = rank;
    c->world = world;
    snprintf(c->name, sizeof(c->name), "%.60s", id.internal);
    const size_t ctrl = (sizeof(Segment) + 4095) & ~(size_t)4095;
    c->map_bytes = ctrl + (size_t)world * world * kSlotBytes;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) return 1;
    if (ftruncate(fd, (off_t)c->map_bytes) != 0) return 1;  // sparse: pages appear when touched
    void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return 1;
    c->seg = static_cast<Segment *>(p);  // a fresh shm object is zero filled: valid initial state of every atomic
    c->data = static_cast<uint8_t *>(p) + ctrl;
    c->seg->arrived.fetch_add(1);
    while (c->seg->arrived.load() < world) usleep(100);  // the collective part of the real call
    *out = c;
    return 0;
}

int ncclCommDestroy(void *comm) {
    Comm *c = static_cast<Comm *>(comm);
    if (!c) return 0;
    munmap(c->seg, c->map_bytes);
    shm_unlink(c->name);  // the last unlink wins; the object lives until every rank has unmapped
    delete c;
    return 0;
}

int ncclGroupStart() {
    ++g_depth;
    return 0;
}

int ncclGroupEnd() {
    if (--g_depth > 0) return 0;
    int rc = 0;
    // sends first: they only wait for their own mailbox to be free, so no rank can block a receive of another
    for (const Op &op : g_ops)
        if (op.send && !rc) rc = run(op);
    for (const Op &op : g_ops)
        if (!op.send && !rc) rc = run(op);
    g_ops.clear();
    return rc;
}

int ncclSend(const void *buf, size_t count, int /*dtype: bytes*/, int peer, void *comm, hipStream_t stream) {
    const Op op = {true, const_cast<void *>(buf), count, peer, static_cast<Comm *>(comm), stream};
    if (g_depth > 0) {
        g_ops.push_back(op);
        return 0;
    }
    return run(op);
}

int ncclRecv(void *buf, size_t count, int, int peer, void *comm, hipStream_t stream) {
    const Op op = {false, buf, count, peer, static_cast<Comm *>(comm), stream};
    if (g_depth > 0) {
        g_ops.push_back(op);
        return 0;
    }
    return run(op);
}

}  // extern "C"
