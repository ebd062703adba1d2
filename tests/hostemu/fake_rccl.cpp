// TEST INFRASTRUCTURE: a stand-in for librccl.so.1 that lets the `not gpu` tests run the REAL tokenizer_amd/csrc/tkz_comm.cpp -- its dlopen,
// its hand-declared prototypes, the id exchange, the argument order and data type of its ncclAllGather, the layout of the gathered table --
// at world sizes 2 and 8 on a CPU, where RCCL itself cannot run (it refuses two ranks on one device, and there is no device here).
// Built as tests/hostemu/_build/fake_rccl/librccl.so.1 and found by tkz_comm.cpp's dlopen("librccl.so.1") through LD_LIBRARY_PATH.
//
// The part of RCCL's C API tkz_comm.cpp binds (rccl.h), over a Unix-domain socket: rank 0 listens on a path carried in the 128-byte unique
// id, the other ranks connect; an all-gather sends every rank's block to rank 0, which sends the table back.  "Device" pointers are host
// pointers in the emulated build, the stream argument is ignored (the emulator's streams are synchronous).
#include <sys/socket.h>
#include <sys/un.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {

struct ncclUniqueId { char internal[128]; };
struct ncclComm { int rank, world; int listen_fd; std::vector<int> fds; char path[108]; };
typedef ncclComm* ncclComm_t;
typedef int ncclResult_t;      // ncclSuccess 0, ncclSystemError 2, ncclInvalidArgument 4 (rccl.h)
typedef int ncclDataType_t;    // ncclInt8 0, ncclUint8 1, ncclInt32 2, ncclUint32 3, ncclInt64 4, ncclUint64 5, ncclFloat16 6, ncclFloat32 7, ncclFloat64 8

static bool send_all(int fd, const void* p, size_t n) {
    const char* c = static_cast<const char*>(p);
    while (n) { const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL); if (k <= 0) { if (errno == EINTR) continue; return false; } c += k; n -= (size_t)k; }
    return true;
}
static bool recv_all(int fd, void* p, size_t n) {
    char* c = static_cast<char*>(p);
    while (n) { const ssize_t k = ::recv(fd, c, n, 0); if (k <= 0) { if (k < 0 && errno == EINTR) continue; return false; } c += k; n -= (size_t)k; }
    return true;
}
static size_t elem_size(ncclDataType_t t) {
    switch (t) { case 0: case 1: return 1; case 2: case 3: case 7: return 4; case 4: case 5: case 8: return 8; case 6: return 2; default: return 0; }
}

ncclResult_t ncclGetVersion(int* v) { if (!v) return 4; *v = 22606; return 0; }      // (what the GPU box's RCCL reports: 2.26.6)
const char* ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : r == 2 ? "unhandled system error (fake rccl)" : r == 4 ? "invalid argument (fake rccl)" : "error (fake rccl)"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return 4;
    memset(id, 0, sizeof *id);
    static int counter = 0;
    snprintf(id->internal, sizeof id->internal, "/tmp/tkz_fake_rccl_%d_%d_%ld", (int)getpid(), counter++, (long)random());
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
    if (!out || world < 1 || rank < 0 || rank >= world) return 4;
    ncclComm* c = new ncclComm();
    c->rank = rank; c->world = world; c->listen_fd = -1;
    id.internal[sizeof id.internal - 1] = 0;
    snprintf(c->path, sizeof c->path, "%.100s", id.internal);
    sockaddr_un addr;
    memset(&addr, 0, sizeof addr);
    addr.sun_family = AF_UNIX;
    snprintf(addr.sun_path, sizeof addr.sun_path, "%.100s", c->path);
    if (rank == 0) {
        c->fds.assign((size_t)world, -1);
        if (world > 1) {
            c->listen_fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
            ::unlink(c->path);
            if (c->listen_fd < 0 || ::bind(c->listen_fd, (sockaddr*)&addr, sizeof addr) != 0 || ::listen(c->listen_fd, world) != 0) { delete c; return 2; }
            for (int k = 1; k < world; ++k) {
                const int fd = ::accept(c->listen_fd, nullptr, nullptr);
                int32_t r = -1;
                if (fd < 0 || !recv_all(fd, &r, 4) || r < 1 || r >= world || c->fds[(size_t)r] >= 0) { delete c; return 2; }
                c->fds[(size_t)r] = fd;
            }
        }
    } else {
        const int fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
        if (fd < 0) { delete c; return 2; }
        int tries = 0;
        while (::connect(fd, (sockaddr*)&addr, sizeof addr) != 0) { if (++tries > 3000) { ::close(fd); delete c; return 2; } usleep(10000); }   // (rank 0 may not be listening yet)
        const int32_t r = rank;
        if (!send_all(fd, &r, 4)) { ::close(fd); delete c; return 2; }
        c->fds.assign(1, fd);
    }
    *out = c;
    return 0;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
    if (!c) return 4;
    for (int fd : c->fds) if (fd >= 0) ::close(fd);
    if (c->listen_fd >= 0) { ::close(c->listen_fd); ::unlink(c->path); }
    delete c;
    return 0;
}
ncclResult_t ncclCommCount(const ncclComm_t c, int* n) { if (!c || !n) return 4; *n = c->world; return 0; }
ncclResult_t ncclCommUserRank(const ncclComm_t c, int* r) { if (!c || !r) return 4; *r = c->rank; return 0; }

// sendcount elements of `type` from every rank; recvbuff = world * sendcount elements, rank r's block at r * sendcount
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t type, ncclComm_t c, void* /*stream*/) {
    const size_t es = elem_size(type);
    if (!c || !sendbuff || !recvbuff || !es) return 4;
    const size_t blk = sendcount * es;
    char* table = static_cast<char*>(recvbuff);
    if (c->rank == 0) {
        memmove(table, sendbuff, blk);
        for (int r = 1; r < c->world; ++r) if (!recv_all(c->fds[(size_t)r], table + (size_t)r * blk, blk)) return 2;
        for (int r = 1; r < c->world; ++r) if (!send_all(c->fds[(size_t)r], table, blk * (size_t)c->world)) return 2;
    } else {
        if (!send_all(c->fds[0], sendbuff, blk) || !recv_all(c->fds[0], table, blk * (size_t)c->world)) return 2;
    }
    return 0;
}

}  // extern "C"
