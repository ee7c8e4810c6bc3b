// comm.hip -- the data-parallel exchange step behind the C-ABI (qst_comm_init / qst_allreduce_bucket).
//
// The reference has no communication at all (single process, training/main.py:113); the data-parallel path this build
// adds exchanges ONE thing, a sum of slices of the fp32 gradient arena, so the boundary is one communicator handle and
// one in-place all-reduce. torch.distributed drives the same RCCL from the Python side (trainer.py); these entry points
// are what a caller WITHOUT torch uses: one process per GPU, rank 0 makes the 128-byte id (qst_comm_unique_id), ships
// it to its peers by whatever it has (a file, MPI, a socket), every rank calls qst_comm_init, and after each stage of
// qst_encoder_backward_stage enqueues qst_allreduce_bucket on its communication stream.
//
// RCCL is bound at run time (dlopen), not at link time: a process that already carries an RCCL (PyTorch bundles its own
// librccl.so) must not get a second copy with a second set of global state -- the loaded one is reused (RTLD_NOLOAD
// first); libqst.so itself keeps no link dependency on it, so single-GPU users need no RCCL at all.
#include <dlfcn.h>
#include <stdint.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>

#include <hip/hip_runtime.h>

#include "qst_common.h"
#include "qst_kernels.h"

namespace {

// the slice of rccl.h this file needs (values from /opt/rocm/include/rccl/rccl.h, stable across NCCL 2.x)
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
enum { kNcclSum = 0, kNcclFloat32 = 7, kNcclBfloat16 = 9 };
typedef int (*GetUniqueIdFn)(NcclUniqueId*);
typedef int (*CommInitRankFn)(NcclComm*, int, NcclUniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, NcclComm, hipStream_t);
typedef int (*CommDestroyFn)(NcclComm);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn error_string = nullptr;
};
Rccl g_rccl;                                  // written once under the mutex, read only after g_rccl_ready
std::atomic<bool> g_rccl_ready{false};
std::mutex g_rccl_mutex;                      // binding happens once, from whichever thread asks first
std::string g_load_error;                     // why the binding failed (under the mutex)
thread_local int t_last_error = 0;            // the calling thread's last RCCL status (as qst_set_hip_error keeps HIP's)

bool rccl_load() {
    if (g_rccl_ready.load(std::memory_order_acquire)) return true;
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl_ready.load(std::memory_order_relaxed)) return true;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;          // the copy this process already has
    if (!h)
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) { g_load_error = "RCCL could not be loaded (librccl.so / librccl.so.1)"; return false; }
    Rccl r;
    r.lib = h;
    r.get_unique_id = (GetUniqueIdFn)dlsym(h, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(h, "ncclCommInitRank");
    r.all_reduce = (AllReduceFn)dlsym(h, "ncclAllReduce");
    r.comm_destroy = (CommDestroyFn)dlsym(h, "ncclCommDestroy");
    r.error_string = (GetErrorStringFn)dlsym(h, "ncclGetErrorString");
    const char* missing = !r.get_unique_id ? "ncclGetUniqueId" : !r.comm_init_rank ? "ncclCommInitRank"
                          : !r.all_reduce ? "ncclAllReduce" : !r.comm_destroy ? "ncclCommDestroy" : nullptr;
    if (missing) {
        g_load_error = std::string("librccl.so is loaded but does not export ") + missing;
        dlclose(h);
        return false;
    }
    g_rccl = r;
    g_rccl_ready.store(true, std::memory_order_release);
    return true;
}

int rccl_rc(int rc) {
    if (rc == 0) return QST_OK;
    t_last_error = rc;
    return QST_ERR_COMM;
}

}  // namespace

struct qst_comm {
    NcclComm comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" int qst_comm_unique_id(void* id_out) {
    if (!id_out) return QST_ERR_BAD_ARG;
    if (!rccl_load()) return QST_ERR_COMM;
    NcclUniqueId id;
    if (int rc = rccl_rc(g_rccl.get_unique_id(&id))) return rc;
    memcpy(id_out, id.internal, QST_COMM_ID_BYTES);
    return QST_OK;
}

extern "C" int qst_comm_init(int rank, int world, const void* unique_id, qst_comm** out) {
    if (!unique_id || !out || world <= 0 || rank < 0 || rank >= world) return QST_ERR_BAD_ARG;
    if (!rccl_load()) return QST_ERR_COMM;
    NcclUniqueId id;
    memcpy(id.internal, unique_id, QST_COMM_ID_BYTES);
    qst_comm* c = new qst_comm;
    c->rank = rank;
    c->world = world;
    if (int rc = rccl_rc(g_rccl.comm_init_rank(&c->comm, world, id, rank))) {      // binds to the current HIP device
        delete c;
        return rc;
    }
    *out = c;
    return QST_OK;
}

extern "C" int qst_allreduce_bucket(qst_comm* comm, void* ptr, int64_t count, int dtype, void* stream) {
    if (!comm || !comm->comm || !ptr || count < 0) return QST_ERR_BAD_ARG;
    if (dtype != QST_COMM_F32 && dtype != QST_COMM_BF16) return QST_ERR_BAD_ARG;
    if (count == 0) return QST_OK;
    return rccl_rc(g_rccl.all_reduce(ptr, ptr, (size_t)count, dtype == QST_COMM_F32 ? kNcclFloat32 : kNcclBfloat16, kNcclSum,
                                     comm->comm, (hipStream_t)stream));
}

extern "C" int qst_comm_rank(const qst_comm* comm) { return comm ? comm->rank : QST_ERR_BAD_ARG; }
extern "C" int qst_comm_world(const qst_comm* comm) { return comm ? comm->world : QST_ERR_BAD_ARG; }

// (A handle is invalid after qst_comm_destroy, as a freed pointer is: the library cannot tell a destroyed handle from a live
// one without keeping a registry; callers drop the handle when they destroy it, as comm.NativeComm does.)
extern "C" void qst_comm_destroy(qst_comm* comm) {
    if (!comm) return;
    if (comm->comm && g_rccl_ready.load(std::memory_order_acquire)) (void)g_rccl.comm_destroy(comm->comm);
    comm->comm = nullptr;
    delete comm;
}

// text for the calling thread's last QST_ERR_COMM
extern "C" const char* qst_comm_last_error(void) {
    if (!g_rccl_ready.load(std::memory_order_acquire)) {
        static thread_local std::string msg;
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        msg = g_load_error.empty() ? "RCCL has not been loaded yet" : g_load_error;
        return msg.c_str();
    }
    if (g_rccl.error_string && t_last_error) return g_rccl.error_string(t_last_error);
    return "";
}
