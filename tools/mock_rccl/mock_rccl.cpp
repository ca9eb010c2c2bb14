// mock_rccl.cpp -- a stand-in for librccl.so.1 that moves bytes between PROCESSES ON ONE GPU through named pipes.
// Test infrastructure only (tests/test_parity_gpu.py::test_cpp_host_three_ranks_over_a_mock_rccl): RCCL refuses two
// ranks on one device, and one device is all a test box has, so the multi-rank logic of csrc/gs_dist.cpp -- which rank
// owns which rows, where a peer's strip lands in the root's buffer, the re-interleaving of rows dealt round-robin --
// could otherwise only run with a world of one.  With this library first on LD_LIBRARY_PATH, gs_dist.cpp's
// dlopen("librccl.so.1") binds the eight entry points below instead: ncclSend = wait for the stream, copy the bytes to
// the host, write them into the pipe <id>_<src>_<dst>; ncclRecv = read them and copy them to the device.
// Nothing of RCCL's performance or protocol is modelled; only the data path gs_dist.cpp programs against.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

// Every pipe is opened O_RDWR once, for the life of the communicator: such an open never blocks, and a writer can never
// meet a pipe without a reader (no SIGPIPE when one message's reader closes while the next message is being written).
struct MockComm { int rank, world; std::string prefix; std::vector<int> to, from; };

static std::string pipe_name(const MockComm* c, int src, int dst) {
    return c->prefix + "_" + std::to_string(src) + "_" + std::to_string(dst);
}
static bool io_all(int fd, char* p, size_t n, bool wr) {
    while (n) {
        const ssize_t k = wr ? write(fd, p, n) : read(fd, p, n);
        if (k <= 0) return false;
        p += k; n -= (size_t)k;
    }
    return true;
}

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    std::memset(id, 0, sizeof(*id));
    const char* dir = std::getenv("MOCK_RCCL_DIR");
    std::snprintf(id->internal, sizeof(id->internal), "%s/mock_%d_%ld", dir ? dir : "/tmp", (int)getpid(), (long)random());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    MockComm* c = new MockComm{rank, nranks, std::string(id.internal), std::vector<int>(nranks, -1), std::vector<int>(nranks, -1)};
    for (int s = 0; s < nranks; ++s)                 // every rank creates every pipe; EEXIST is fine
        for (int d = 0; d < nranks; ++d)
            if (s != d) (void)mkfifo(pipe_name(c, s, d).c_str(), 0600);
    for (int p = 0; p < nranks; ++p) {
        if (p == rank) continue;
        c->to[p] = open(pipe_name(c, rank, p).c_str(), O_RDWR);
        c->from[p] = open(pipe_name(c, p, rank).c_str(), O_RDWR);
        if (c->to[p] < 0 || c->from[p] < 0) { delete c; return ncclSystemError; }
    }
    *comm = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    for (int p = 0; p < c->world; ++p)
        if (p != c->rank) {
            if (c->to[p] >= 0) close(c->to[p]);
            if (c->from[p] >= 0) close(c->from[p]);
            (void)unlink(pipe_name(c, c->rank, p).c_str()); (void)unlink(pipe_name(c, p, c->rank).c_str());
        }
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    if (type != ncclUint8 && type != ncclChar) return ncclInvalidArgument;
    std::vector<char> host(count);
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(host.data(), sendbuff, count, hipMemcpyDeviceToHost) != hipSuccess)
        return ncclUnhandledCudaError;
    if (peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    return io_all(c->to[peer], host.data(), count, true) ? ncclSuccess : ncclSystemError;       // blocks while the pipe is full
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    MockComm* c = reinterpret_cast<MockComm*>(comm);
    if (type != ncclUint8 && type != ncclChar) return ncclInvalidArgument;
    std::vector<char> host(count);
    if (peer < 0 || peer >= c->world || peer == c->rank) return ncclInvalidArgument;
    if (!io_all(c->from[peer], host.data(), count, false)) return ncclSystemError;
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(recvbuff, host.data(), count, hipMemcpyHostToDevice) != hipSuccess)
        return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "mock rccl error"; }

}  // extern "C"
