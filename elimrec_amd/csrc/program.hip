// A training step as ONE host call: a recorded list of C-ABI calls, stream hand-overs and RCCL collectives, issued from C.
//
// The step of shard.py is some 13 kernel launches of 15-60 us on two HIP streams; issued call by call from Python (even
// from pre-converted ctypes argument lists) the host needs 0.2 ms per 0.32 ms step and the multi-rank step, with four
// torch.distributed collectives on top, is host-bound. A `program` is what such a step does, written down once:
//   CALL        one entry point of this library with its arguments packed as 64-bit words (pointers, integers, fp32 bits)
//   RECORD/WAIT hipEventRecord on one stream / hipStreamWaitEvent on another: the fork and join of the second stream
// The step's collectives are entry points of this library too (elimrec_comm_*: ncclAllGather / ncclAllReduce / grouped
// ncclSend + ncclRecv on a communicator the library owns -- RCCL over xGMI, enqueued on the caller's stream, no host round
// trip, no hand-over to another library's internal stream), so a program lists them like any kernel launch.
// elimrec_program_run walks the list; `patches` overwrite argument words that change from step to step (the batch's index
// tensors, the loss slot, Adam's step count) before the walk. The Python side (elimrec_amd/program.py) builds a program by
// tracing steps of the ordinary path and diffing their calls, so a program issues exactly what that path issues.
#include "common.h"
#include <dlfcn.h>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

namespace elimrec {

template <typename T> struct Unpack {
    static T get(uint64_t v) {
        if constexpr (std::is_pointer<T>::value) return reinterpret_cast<T>(static_cast<uintptr_t>(v));
        else if constexpr (std::is_same<T, float>::value) { uint32_t b = (uint32_t)v; float f; memcpy(&f, &b, 4); return f; }
        else return static_cast<T>(v);
    }
};
template <typename R, typename... A, size_t... I>
static R call_packed_impl(R (*f)(A...), const uint64_t *a, std::index_sequence<I...>) { return f(Unpack<A>::get(a[I])...); }
template <typename R, typename... A> static R call_packed(R (*f)(A...), const uint64_t *a) {
    static_assert(sizeof...(A) <= ELIMREC_PROGRAM_MAX_ARGS, "too many arguments for a program op");
    return call_packed_impl(f, a, std::index_sequence_for<A...>{});
}
template <typename R, typename... A> constexpr int arg_count(R (*)(A...)) { return (int)sizeof...(A); }

// The events of a program WITHOUT collectives order streams of ONE device whose buffers no other device writes: no system-scope
// fence (cache write-back + invalidate for the host and other devices) when one is recorded -- the kernels' own agent-scope
// release / acquire is what the streams need of each other. A program that lists the library's RCCL calls keeps the default
// events: peers write into this device's buffers over xGMI, and a stream that consumes them behind an event must see them at
// system scope (elimrec_program_create_scoped states it).
static unsigned event_flags(bool has_collectives) {
    return has_collectives ? hipEventDisableTiming : (hipEventDisableTiming | hipEventDisableSystemFence);
}

struct FnEntry { const char *name; int (*thunk)(const uint64_t *); int n_args; };
#define ELIMREC_FN(fn) {#fn, [](const uint64_t *a) -> int { return call_packed(fn, a); }, arg_count(fn)}
static const FnEntry kFns[] = {
    ELIMREC_FN(elimrec_batch_plan), ELIMREC_FN(elimrec_slab_hop), ELIMREC_FN(elimrec_slab_sweep_hop), ELIMREC_FN(elimrec_slab_sweep_hop_adam), ELIMREC_FN(elimrec_slab_source_bits), ELIMREC_FN(elimrec_slab_hop_bwd_w),
    ELIMREC_FN(elimrec_slab_hop_adam), ELIMREC_FN(elimrec_slab_rows), ELIMREC_FN(elimrec_slab_merge_rows),
    ELIMREC_FN(elimrec_head_fwd_fused), ELIMREC_FN(elimrec_bpr_head_rows),
    ELIMREC_FN(elimrec_bpr_head_rows_sum), ELIMREC_FN(elimrec_bpr_head_rows_sum_pub), ELIMREC_FN(elimrec_sum), ELIMREC_FN(elimrec_segment_apply_head_bwd),
    ELIMREC_FN(elimrec_segment_apply_head_bwd_packed), ELIMREC_FN(elimrec_segment_apply_head_bwd_sources),
    ELIMREC_FN(elimrec_segment_apply_head_bwd_split), ELIMREC_FN(elimrec_linear_fwd_batched), ELIMREC_FN(elimrec_linear_bwd_w_batched),
    ELIMREC_FN(elimrec_linear_bwd_w_batched_merge), ELIMREC_FN(elimrec_linear_bwd_w_reduce), ELIMREC_FN(elimrec_adam_multi),
    ELIMREC_FN(elimrec_source_rows_split), ELIMREC_FN(elimrec_copy_cols), ELIMREC_FN(elimrec_lookup_pack), ELIMREC_FN(elimrec_lookup_unpack),
    ELIMREC_FN(elimrec_lookup_counts), ELIMREC_FN(elimrec_adam_step_out),
    ELIMREC_FN(elimrec_peer_cols_to_rows), ELIMREC_FN(elimrec_rows_bitmap), ELIMREC_FN(elimrec_comm_all_gather), ELIMREC_FN(elimrec_comm_all_reduce_f32),
    ELIMREC_FN(elimrec_comm_all_to_all), ELIMREC_FN(elimrec_comm_all_to_all_v), ELIMREC_FN(elimrec_wide_from_master),
    ELIMREC_FN(elimrec_wide_rows), ELIMREC_FN(elimrec_wide_grad), ELIMREC_FN(elimrec_head_fwd_fused_rows), ELIMREC_FN(elimrec_head_fwd_fused_src16),
    ELIMREC_FN(elimrec_head_fwd_fused_peers),
};
constexpr int kNumFns = (int)(sizeof(kFns) / sizeof(kFns[0]));

// ---- RCCL, resolved at run time from the librccl this process already holds (PyTorch-ROCm loads one; no second copy)
typedef struct ncclComm *ncclComm_t;
struct NcclId { char internal[128]; };
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(NcclId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, NcclId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(ncclComm_t, int *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
static Rccl g_rccl;
static int rccl_load() {
    if (g_rccl.h) return 0;
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);          // the copy already mapped (torch/lib/librccl.so)
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { set_error("program: librccl.so not found (%s)", dlerror()); return ELIMREC_E_UNSUPPORTED; }
#define ELIMREC_SYM(field, name)                                                         \
    *(void **)(&g_rccl.field) = dlsym(h, name);                                          \
    if (!g_rccl.field) { set_error("program: %s missing in librccl", name); return ELIMREC_E_UNSUPPORTED; }
    ELIMREC_SYM(GetUniqueId, "ncclGetUniqueId") ELIMREC_SYM(CommInitRank, "ncclCommInitRank") ELIMREC_SYM(CommDestroy, "ncclCommDestroy")
    ELIMREC_SYM(CommCount, "ncclCommCount")
    ELIMREC_SYM(AllGather, "ncclAllGather") ELIMREC_SYM(AllReduce, "ncclAllReduce") ELIMREC_SYM(Send, "ncclSend") ELIMREC_SYM(Recv, "ncclRecv")
    ELIMREC_SYM(GroupStart, "ncclGroupStart") ELIMREC_SYM(GroupEnd, "ncclGroupEnd") ELIMREC_SYM(GetErrorString, "ncclGetErrorString")
#undef ELIMREC_SYM
    g_rccl.h = h;
    return 0;
}
static int rccl_check(int rc, const char *what) {
    if (rc == 0) return 0;
    set_error("%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error");
    return 20000 + rc;
}
constexpr int kNcclFloat32 = 7, kNcclUint8 = 1, kNcclSum = 0;     // ncclDataType_t / ncclRedOp_t values (nccl.h)

struct Comm { ncclComm_t comm; int world, rank; };

struct Program {
    std::vector<elimrec_op> ops;
    std::vector<hipEvent_t> events;
};

}  // namespace elimrec

using namespace elimrec;

extern "C" int elimrec_program_fn_count(void) { return kNumFns; }
extern "C" const char *elimrec_program_fn_name(int i) { return (i >= 0 && i < kNumFns) ? kFns[i].name : nullptr; }
extern "C" int elimrec_program_fn_args(int i) { return (i >= 0 && i < kNumFns) ? kFns[i].n_args : -1; }

extern "C" int elimrec_comm_unique_id(void *id128) {
    int rc = rccl_load();
    if (rc) return rc;
    return rccl_check(g_rccl.GetUniqueId((NcclId *)id128), "ncclGetUniqueId");
}

extern "C" int elimrec_comm_create(const void *id128, int world, int rank, void **comm_out) {
    ELIMREC_REQUIRE(id128 && comm_out && world >= 1 && rank >= 0 && rank < world, "comm_create: bad arguments");
    int rc = rccl_load();
    if (rc) return rc;
    NcclId id;
    memcpy(&id, id128, sizeof(id));
    Comm *c = new Comm{nullptr, world, rank};
    rc = rccl_check(g_rccl.CommInitRank(&c->comm, world, id, rank), "ncclCommInitRank");
    if (rc) { delete c; return rc; }
    *comm_out = c;
    return 0;
}

extern "C" int elimrec_comm_nranks(void *comm, int *n_out) {
    ELIMREC_REQUIRE(comm && n_out, "comm_nranks: bad arguments");
    return rccl_check(g_rccl.CommCount(((Comm *)comm)->comm, n_out), "ncclCommCount");
}

extern "C" int elimrec_comm_destroy(void *comm) {
    if (!comm) return 0;
    Comm *c = (Comm *)comm;
    int rc = g_rccl.CommDestroy ? rccl_check(g_rccl.CommDestroy(c->comm), "ncclCommDestroy") : 0;
    delete c;
    return rc;
}

// Collectives on the library's communicator, enqueued on `stream` (no host synchronisation, no internal stream): what the
// step's exchanges are made of. They are ordinary entry points, so a program lists them like any kernel launch.
extern "C" int elimrec_comm_all_gather(void *comm, const void *d_send, void *d_recv, int64_t bytes_per_rank, void *stream) {
    ELIMREC_REQUIRE(comm && d_send && d_recv && bytes_per_rank >= 0, "comm_all_gather: bad arguments");
    Comm *c = (Comm *)comm;
    return rccl_check(g_rccl.AllGather(d_send, d_recv, (size_t)bytes_per_rank, kNcclUint8, c->comm, (hipStream_t)stream), "ncclAllGather");
}

extern "C" int elimrec_comm_all_reduce_f32(void *comm, float *d_buf, int64_t n, void *stream) {
    ELIMREC_REQUIRE(comm && d_buf && n >= 0, "comm_all_reduce: bad arguments");
    Comm *c = (Comm *)comm;
    return rccl_check(g_rccl.AllReduce(d_buf, d_buf, (size_t)n, kNcclFloat32, kNcclSum, c->comm, (hipStream_t)stream), "ncclAllReduce");
}

extern "C" int elimrec_comm_all_to_all(void *comm, const void *d_send, void *d_recv, int64_t bytes_per_peer, void *stream) {
    ELIMREC_REQUIRE(comm && d_send && d_recv && bytes_per_peer >= 0, "comm_all_to_all: bad arguments");
    Comm *c = (Comm *)comm;
    const char *s = (const char *)d_send;
    char *r = (char *)d_recv;
    const size_t nb = (size_t)bytes_per_peer;
    int rc = rccl_check(g_rccl.GroupStart(), "ncclGroupStart");
    for (int q = 0; q < c->world && !rc; ++q) {
        rc = rccl_check(g_rccl.Send(s + (size_t)q * nb, nb, kNcclUint8, q, c->comm, (hipStream_t)stream), "ncclSend");
        if (!rc) rc = rccl_check(g_rccl.Recv(r + (size_t)q * nb, nb, kNcclUint8, q, c->comm, (hipStream_t)stream), "ncclRecv");
    }
    const int rc2 = rccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
    return rc ? rc : rc2;
}

// sizes: HOST int64 [2 x world] = bytes to send to peer 0.., then bytes to receive from peer 0.. (chunks back to back)
extern "C" int elimrec_comm_all_to_all_v(void *comm, const void *d_send, void *d_recv, const int64_t *sizes, void *stream) {
    ELIMREC_REQUIRE(comm && d_send && d_recv && sizes, "comm_all_to_all_v: bad arguments");
    Comm *c = (Comm *)comm;
    const char *s = (const char *)d_send;
    char *r = (char *)d_recv;
    int rc = rccl_check(g_rccl.GroupStart(), "ncclGroupStart");
    size_t so = 0, ro = 0;
    for (int q = 0; q < c->world && !rc; ++q) {
        const size_t sb = (size_t)sizes[q], rb = (size_t)sizes[c->world + q];
        if (sb) rc = rccl_check(g_rccl.Send(s + so, sb, kNcclUint8, q, c->comm, (hipStream_t)stream), "ncclSend");
        if (!rc && rb) rc = rccl_check(g_rccl.Recv(r + ro, rb, kNcclUint8, q, c->comm, (hipStream_t)stream), "ncclRecv");
        so += sb; ro += rb;
    }
    const int rc2 = rccl_check(g_rccl.GroupEnd(), "ncclGroupEnd");
    return rc ? rc : rc2;
}

extern "C" int elimrec_program_create_scoped(const elimrec_op *ops, int n_ops, int system_scope_events, void **prog_out) {
    ELIMREC_REQUIRE(ops && n_ops > 0 && prog_out, "program_create: bad arguments");
    Program *p = new Program();
    p->ops.assign(ops, ops + n_ops);
    int n_events = 0;
    // system-scope events: what the caller says (peers or the host write buffers the program's kernels read behind an event), or --
    // a caller that says nothing -- whenever the list holds one of this library's RCCL calls
    bool has_collectives = system_scope_events != 0;
    for (const elimrec_op &o : p->ops) {
        if (o.kind == ELIMREC_OP_CALL) {
            if (o.fn < 0 || o.fn >= kNumFns) { delete p; set_error("program_create: unknown function index %d", o.fn); return ELIMREC_E_BADARG; }
            has_collectives = has_collectives || strncmp(kFns[o.fn].name, "elimrec_comm_", 13) == 0;
        } else if (o.kind == ELIMREC_OP_RECORD || o.kind == ELIMREC_OP_WAIT) {
            if ((int)o.args[1] + 1 > n_events) n_events = (int)o.args[1] + 1;
        } else {
            delete p; set_error("program_create: unknown op kind %d", o.kind); return ELIMREC_E_BADARG;
        }
    }
    p->events.resize(n_events);
    for (int e = 0; e < n_events; ++e) {
        int rc = check_hip(hipEventCreateWithFlags(&p->events[e], event_flags(has_collectives)), "hipEventCreate");
        if (rc) { delete p; return rc; }
    }
    *prog_out = p;
    return 0;
}

extern "C" int elimrec_program_create(const elimrec_op *ops, int n_ops, void **prog_out) {
    return elimrec_program_create_scoped(ops, n_ops, 0, prog_out);
}

extern "C" int elimrec_program_destroy(void *prog) {
    if (!prog) return 0;
    Program *p = (Program *)prog;
    for (hipEvent_t e : p->events) (void)hipEventDestroy(e);
    delete p;
    return 0;
}

extern "C" int elimrec_program_run(void *prog, const elimrec_patch *patches, int n_patches) {
    ELIMREC_REQUIRE(prog && (n_patches == 0 || patches), "program_run: bad arguments");
    Program *p = (Program *)prog;
    const int n = (int)p->ops.size();
    for (int k = 0; k < n_patches; ++k) {
        ELIMREC_REQUIRE(patches[k].op >= 0 && patches[k].op < n && patches[k].arg >= 0 && patches[k].arg < ELIMREC_PROGRAM_MAX_ARGS,
                        "program_run: patch %d out of range", k);
        p->ops[patches[k].op].args[patches[k].arg] = patches[k].value;
    }
    for (int i = 0; i < n; ++i) {
        const elimrec_op &o = p->ops[i];
        const uint64_t *a = o.args;
        int rc = 0;
        switch (o.kind) {
            case ELIMREC_OP_CALL: rc = kFns[o.fn].thunk(a); break;
            case ELIMREC_OP_RECORD: rc = check_hip(hipEventRecord(p->events[a[1]], (hipStream_t)a[0]), "hipEventRecord"); break;
            case ELIMREC_OP_WAIT: rc = check_hip(hipStreamWaitEvent((hipStream_t)a[0], p->events[a[1]], 0), "hipStreamWaitEvent"); break;
            default: break;
        }
        if (rc) return rc;
    }
    return 0;
}
