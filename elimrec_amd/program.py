"""A training step as ONE host call (csrc/program.hip): tracing the ordinary step, finding what changes from step to step,
building the native program.

The ordinary step (shard.py) issues its launches through the ctypes binding, region by region, with Python in between:
0.2 ms of host time per 0.32 ms step. `StepTracer` records every C-ABI call of a step together with the stream hand-overs;
two traces of steps of the same buffer parity are diffed word by word: arguments that differ must be one of the known
per-step values (the batch's three index tensors, the loss slot, Adam's step count) or the program is not built and the
ordinary path stays. The program issues exactly the calls the traced step issued.
"""
import ctypes
import struct

from . import _lib

OP_CALL, OP_RECORD, OP_WAIT = range(3)


def _fn_table():
    lib = _lib.load()
    return {lib.elimrec_program_fn_name(i).decode(): (i, lib.elimrec_program_fn_args(i)) for i in range(lib.elimrec_program_fn_count())}


def _word(value, ctype):
    """One argument as the 64-bit word program.hip unpacks."""
    if value is None:
        return 0
    if ctype is ctypes.c_float:
        return struct.unpack("<I", struct.pack("<f", float(value)))[0]
    if ctype in (ctypes.c_double,):
        raise TypeError("double arguments are not packable")
    if isinstance(value, int):
        return value & 0xFFFFFFFFFFFFFFFF
    if isinstance(value, float):
        return struct.unpack("<I", struct.pack("<f", value))[0]
    obj = getattr(value, "_obj", None)                 # ctypes.byref(x)
    if obj is not None:
        return ctypes.addressof(obj)
    if isinstance(value, (ctypes.Array, ctypes.Structure)):
        return ctypes.addressof(value)
    if isinstance(value, ctypes._Pointer):
        return ctypes.cast(value, ctypes.c_void_p).value or 0
    if isinstance(value, ctypes._SimpleCData):
        v = value.value
        return 0 if v is None else (_word(v, ctype))
    raise TypeError("cannot pack argument %r" % (value,))


class StepTracer(object):
    """Context: every C-ABI call issued through the binding, and every stream hand-over reported with sync(), in order."""

    def __init__(self):
        self.items = []

    def __enter__(self):
        lib = _lib.load()
        self._prev = lib._trace
        lib._trace = self.items
        return self

    def __exit__(self, *exc):
        _lib.load()._trace = self._prev
        return False


def sync(waiter, signaller):
    """waiter.wait_stream(signaller), reported to an active tracer (torch streams)."""
    waiter.wait_stream(signaller)
    tr = _lib.load()._trace
    if tr is not None:
        tr.append(("sync", int(waiter.cuda_stream), int(signaller.cuda_stream)))


class Recorded(object):
    """An event recorded on a stream at one point of the step (record()), for a wait() issued LATER from another stream: unlike
    sync(), work enqueued on the signalling stream between the two is not waited for."""

    def __init__(self, event, stream, token, trace=None):
        self.event, self.stream, self.token, self.trace = event, stream, token, trace


def record(stream):
    """Record an event on `stream` now; reported to an active tracer as ("record", stream, n-th record of this trace)."""
    import torch
    ev = torch.cuda.Event()
    ev.record(stream)
    tr = _lib.load()._trace
    token = None
    if tr is not None:
        token = sum(1 for it in tr if it[0] == "record")
        tr.append(("record", int(stream.cuda_stream), token))
    return Recorded(ev, stream, token, tr)


def wait(waiter, recorded):
    """waiter waits for the event of record(); reported to an active tracer."""
    waiter.wait_event(recorded.event)
    tr = _lib.load()._trace
    if tr is not None:
        if recorded.token is None or recorded.trace is not tr:
            raise RuntimeError("program.wait: the event was recorded outside this trace")
        tr.append(("wait", int(waiter.cuda_stream), recorded.token))


def words_of(trace):
    """[(kind, name, [words])] of a trace: calls with packed arguments, syncs as (waiter, signaller)."""
    out = []
    for it in trace:
        if it[0] == "sync":
            out.append(("sync", None, [it[1], it[2]]))
        elif it[0] in ("record", "wait"):
            out.append((it[0], None, [it[1], it[2]]))
        else:
            _, name, fn, args = it
            types = fn.argtypes
            out.append(("call", name, [_word(a, t) for a, t in zip(args, types)]))
    return out


class StepProgram(object):
    """A built program + the patch slots of its per-step values. keep: Python objects the packed pointers refer to."""

    def __init__(self, items, varying, keep, system_scope_events=False):
        """items: words_of(trace); varying: {(op index, arg index): label}. system_scope_events: the step's kernels read, behind
        one of the program's events, memory that peers or the host write (a multi-rank step): events keep their system-scope
        fence; otherwise the hand-overs between this device's streams are fence-free."""
        lib = _lib.load()
        table = _fn_table()
        ops, index_of = [], {}
        n_events = sum(1 for kind, _, _ in items if kind == "record")      # events 0 .. of record() / wait(), then one per sync()
        for k, (kind, name, words) in enumerate(items):
            if kind in ("record", "wait"):
                op = _lib.ProgramOp()
                op.kind, op.fn = (OP_RECORD if kind == "record" else OP_WAIT), 0
                op.args[0], op.args[1] = words[0], words[1]
                ops.append(op)
                continue
            if kind == "sync":
                waiter, signaller = words
                rec, wait = _lib.ProgramOp(), _lib.ProgramOp()
                rec.kind, rec.fn = OP_RECORD, 0
                rec.args[0], rec.args[1] = signaller, n_events
                wait.kind, wait.fn = OP_WAIT, 0
                wait.args[0], wait.args[1] = waiter, n_events
                n_events += 1
                ops += [rec, wait]
                continue
            if name not in table:
                raise KeyError("program: %s is not in the native function table" % name)
            fi, n_args = table[name]
            if n_args != len(words):
                raise ValueError("program: %s takes %d arguments, the trace has %d" % (name, n_args, len(words)))
            op = _lib.ProgramOp()
            op.kind, op.fn = OP_CALL, fi
            for j, w in enumerate(words):
                op.args[j] = w
            index_of[k] = len(ops)
            ops.append(op)
        self.n_ops = len(ops)
        self._ops = (_lib.ProgramOp * len(ops))(*ops)
        self._keep = keep
        self._items, self._varying = items, dict(varying)
        self.slots = {}                                  # label -> [(op, arg)]
        for (k, j), label in varying.items():
            self.slots.setdefault(label, []).append((index_of[k], j))
        self._patch_order = sorted(self.slots)
        n_patch = sum(len(v) for v in self.slots.values())
        self._patches = (_lib.ProgramPatch * max(n_patch, 1))()
        at = 0
        self._patch_at = {}
        for label in self._patch_order:
            self._patch_at[label] = []
            for (o, j) in self.slots[label]:
                self._patches[at].op, self._patches[at].arg = o, j
                self._patch_at[label].append(at)
                at += 1
        self._n_patch = n_patch
        handle = ctypes.c_void_p()
        _lib.check(lib.elimrec_program_create_scoped(self._ops, len(ops), 1 if system_scope_events else 0, ctypes.byref(handle)),
                   "program_create")
        self._handle = handle
        self._run = lib.elimrec_program_run

    def run(self, values):
        """values: {label: 64-bit word} for every patch slot."""
        p = self._patches
        for label, ats in self._patch_at.items():
            v = values[label]
            for at in ats:
                p[at].value = v
        rc = self._run(self._handle, p, self._n_patch)
        if rc:
            _lib.check(rc, "program_run")

    def verify(self, trace, known):
        """A freshly traced ordinary step against this program: the same calls in the same order, every argument word either
        the one the program holds or -- at a patch slot -- this step's value of that label (known: {label: word}); host-side
        argument blocks may live elsewhere but must hold the same bytes. Returns None, or what differs (the program freezes
        every argument that did not change between its two traced steps: anything a step starts to vary later shows here)."""
        mine, now = self._items, words_of(trace)
        if len(mine) != len(now):
            return "the step now issues %d items, the program holds %d" % (len(now), len(mine))
        ref = self._keep[0] if self._keep else None
        for k, ((ka, na, xa), (kb, nb, xb)) in enumerate(zip(mine, now)):
            if ka != kb or na != nb or len(xa) != len(xb):
                return "item %d is %s %s now, %s %s in the program" % (k, kb, nb, ka, na)
            for j, (u, v) in enumerate(zip(xa, xb)):
                label = self._varying.get((k, j))
                if label is not None:
                    if known.get(label) != v:
                        return "argument %d of %s (patched as '%s') is %#x, the step's value of that label is %#x" % (j, na, label, v, known.get(label, 0))
                    continue
                if u == v:
                    continue
                if ka == "call" and ref is not None:
                    ha, hb = _host_bytes(ref[k][3][j]), _host_bytes(trace[k][3][j])
                    if ha is not None and ha == hb:
                        continue
                return "argument %d of %s is %#x now, frozen as %#x in the program" % (j, na if na else ka, v, u)
        return None

    def __del__(self):
        try:
            if self._handle:
                _lib.load().elimrec_program_destroy(self._handle)
        except Exception:
            pass


def _host_bytes(value):
    """Contents of a host-side ctypes argument (array, struct, byref), or None for plain words."""
    obj = getattr(value, "_obj", None)
    if obj is not None:
        value = obj
    if isinstance(value, (ctypes.Array, ctypes.Structure)):
        return ctypes.string_at(ctypes.addressof(value), ctypes.sizeof(value))
    return None


def diff_traces(a, b, known_a, known_b):
    """Two traces of the same step structure -> {(op, arg): label} of the words that differ, each explained by one of the
    per-step values (known_x: {label: word} of trace x). Raises ValueError when the structures differ or a difference is not
    one of the known values."""
    wa, wb = words_of(a), words_of(b)
    if len(wa) != len(wb):
        raise ValueError("traces differ in length (%d vs %d)" % (len(wa), len(wb)))
    varying = {}
    for k, ((ka, na, xa), (kb, nb, xb)) in enumerate(zip(wa, wb)):
        if ka != kb or na != nb or len(xa) != len(xb):
            raise ValueError("traces differ in structure at item %d (%s vs %s)" % (k, na, nb))
        for j, (u, v) in enumerate(zip(xa, xb)):
            if u == v:
                continue
            if ka in ("sync", "record", "wait"):
                raise ValueError("stream handles or event numbers differ between the traces")
            label = next((lab for lab in known_a if known_a[lab] == u and known_b[lab] == v), None)
            if label is not None:
                varying[(k, j)] = label
                continue
            ha, hb = _host_bytes(a[k][3][j]), _host_bytes(b[k][3][j])
            if ha is not None and ha == hb:
                continue            # a host-side argument block rebuilt per call with the same contents: trace a's copy is kept alive
            label = next((lab for lab in known_a if known_a[lab] == u and known_b[lab] == v), None)
            if label is None:
                raise ValueError("argument %d of %s changes between steps (%#x -> %#x) and is none of %s" % (j, na, u, v, sorted(known_a)))
            varying[(k, j)] = label
    return wa, varying
