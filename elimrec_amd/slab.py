"""Slab-major (column-sharded) propagation: host plan + tensor-level wrappers of the elimrec_slab_* entry points.

A table of `dl` columns is kept as `ns` slabs of width `w` floats (include/elimrec_hip.h, "slab-major propagation"):
`SlabTable.data` is one flat fp32 tensor [ns * n * w]; slab s holds columns [s*w, (s+1)*w) of every row. The
adjacency (models/EliMRec.py:309-354, as built by model.create_adj_mat) is turned ONCE on the host into the SELL-64
work-item form the hop kernel consumes (`SellPlan`).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from .ops import _dev, _stream

# rows with more non-zeros are cut into segments of this length: the longest work item bounds a hop's critical path
# (32: 21 us per hop on an 8-column shard at the Tiktok shape; 64: 29 us; 16: 40 us -- too many partial rows)
LONG_ROW_THRESHOLD = 32
SLAB_W_CAP = 32      # widest slab (floats): a 128-B row piece per gather (tools/hop_only.py overrides it for geometry sweeps)
SLAB_GROUPS = 0      # 0: as many slab groups as divide the slab count and the 8 XCDs


def choose_slabs(dl, n_rows=None):
    """(ns, w) for a table of dl columns: w = the largest power-of-two multiple of 4 dividing dl, capped at 32 floats.
    Measured at the Tiktok shape (round 2; d = 64, us per hop): w = 64 (row-major) 44, w = 32 in two
    slab groups 35, w = 16 in four 42, w = 8 in eight 67 -- a gathered piece narrower than one 128-B cache line still
    moves a whole line from L2 to the CU, so narrower slabs lose more on the L2 -> L1 path than their smaller L2
    footprint wins; 128-B pieces halve every XCD's footprint at no cost per line."""
    if dl % 4 != 0:
        raise ValueError("column count must be a multiple of 4 (got %d)" % dl)
    cap = SLAB_W_CAP
    w = 4
    while w * 2 <= cap and dl % (w * 2) == 0:
        w *= 2
    return dl // w, w


def choose_groups(ns):
    """Slab groups of a hop launch: as many as divide both the slab count and the 8 XCDs."""
    gs = SLAB_GROUPS
    if gs > 0 and ns % gs == 0:
        return gs
    for gs in (8, 4, 2, 1):
        if ns % gs == 0 and _is_pow2(ns // gs):
            return gs
    return ns


def _is_pow2(x):
    return x >= 1 and (x & (x - 1)) == 0


class SellPlan(object):
    """SELL-64 work items of a CSR matrix on a device (struct elimrec_sell)."""

    def __init__(self, m, device, threshold=LONG_ROW_THRESHOLD, side_split=None, tiered=None, ipw=8, phase=None, rows_from=0, share=None):
        """side_split = U: the unsplit rows are processed side by side (item rows, then user rows; by decreasing
        length inside a side) instead of by length alone -- all workgroups then gather from the same side's rows at
        the same time, which is what an XCD's L2 can hold (users read items by popularity, items read the whole,
        smaller, user table).
        tiered (the engine turns it on for whole fp32 tables, where it measured faster): rows above `threshold` are not cut into segments
        but given to one wave (up to 64 neighbours per lane group at ipw <= 8, else 32) or one workgroup (up to 64 per lane group:
        256*ipw) each, and only the rows longer still are segmented -- one launch per hop, no fix-up launch. ipw = lane
        groups per wave of the geometry the plan will mostly run with (64 / lanes per work item)."""
        if tiered is None:
            tiered = False
        m = m.tocsr()
        m.sort_indices()
        n_rows, n_src = m.shape
        if max(n_rows, n_src) >= 2 ** 31 - 1:
            raise ValueError("node ids must fit int32 (%d x %d matrix)" % (n_rows, n_src))
        # offsets into the index arrays (tile_off, rowptr) are int64: >= 2^31 non-zeros build (BASELINE.json configs[4]: 2e9)
        rowptr = m.indptr.astype(np.int64)
        col = m.indices.astype(np.int32)
        val = m.data.astype(np.float32)
        deg = np.diff(rowptr)
        T = int(threshold)
        # a wave per row up to 64 neighbours per lane group at 8 groups (measured: 0.319 against 0.322 ms per step with 32),
        # 32 with more, narrower groups (column shards; not re-measured)
        T1 = (64 if ipw <= 8 else 32) * ipw if tiered else T
        T2 = 256 * ipw if tiered else T
        long_rows = np.nonzero(deg > T)[0]
        short_rows = np.nonzero(deg <= T)[0]
        if rows_from:                # a plan of rows [rows_from, n) only: the rows below belong to another launch (SweepPlan)
            long_rows, short_rows = long_rows[long_rows >= rows_from], short_rows[short_rows >= rows_from]
        self.sweep = None
        w1 = long_rows[deg[long_rows] <= T1]
        w4 = long_rows[(deg[long_rows] > T1) & (deg[long_rows] <= T2)]
        w1 = w1[np.argsort(-deg[w1], kind="stable")]
        w4 = w4[np.argsort(-deg[w4], kind="stable")]
        # segments of the split rows (tiered: of the rows above T2 only): slot numbers run row by row, segment by segment
        TS = T                                                                 # segment length of the rows above T2
        nseg = np.where(deg[long_rows] > T2, (deg[long_rows] + TS - 1) // TS, 0) if tiered else (deg[long_rows] + T - 1) // T
        seg_ptr = np.concatenate([[0], np.cumsum(nseg)]).astype(np.int64)
        n_seg = int(seg_ptr[-1])
        seg_row = np.repeat(np.arange(len(long_rows)), nseg)
        k = np.arange(n_seg) - seg_ptr[seg_row]
        seg_beg = rowptr[long_rows][seg_row] + k * TS
        seg_len = np.minimum(TS, rowptr[long_rows + 1][seg_row] - seg_beg)
        # heavy first, inside each kind
        so = np.argsort(-seg_len, kind="stable")
        if phase is not None:        # explicit processing phases of the unsplit rows (int per row), by decreasing length inside
            ro = np.lexsort((-deg[short_rows], np.asarray(phase)[short_rows]))
        elif side_split is None:
            ro = np.argsort(-deg[short_rows], kind="stable")
        else:
            ro = np.lexsort((-deg[short_rows], short_rows < int(side_split)))
        long_index = np.full(n_rows, -1, np.int32)
        long_index[long_rows] = np.arange(len(long_rows), dtype=np.int32)
        dev = torch.device(device)
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(np.asarray(a).astype(dt))).to(dev)
        self.device = dev
        self.n_rows, self.n_src, self.nnz = int(n_rows), int(n_src), int(m.nnz)
        self.n_seg, self.n_long = n_seg, int(len(long_rows))
        self.threshold = T
        self.tiered, self.n_w1, self.n_w4 = bool(tiered), int(len(w1)), int(len(w4))
        self.t = dict(long_rows=t(long_rows if len(long_rows) else np.zeros(1), np.int32),
                      long_seg_ptr=t(seg_ptr, np.int32), long_index=t(long_index, np.int32))
        if share is not None:        # a plan of some rows of the SAME matrix as `share` (SweepPlan.items): one CSR copy on the device
            self.t.update({k: share.t[k] for k in ("rowptr", "csr_col", "csr_val")})
            self.shared = ("rowptr", "csr_col", "csr_val")
        else:
            self.t.update(rowptr=t(rowptr, np.int64), csr_col=t(col if len(col) else np.zeros(1), np.int32),
                          csr_val=t(val if len(val) else np.zeros(1), np.float32))
            self.shared = ()
        p = lambda k: self.t[k].data_ptr() if k in self.t else None
        if tiered:
            tiles = self._wave_tiles(int(ipw), rowptr, col, val, deg, w4, w1, seg_beg[so], seg_len[so], so, seg_row[so], short_rows[ro])
            self.t.update({k: t(v, np.float32 if k == "tile_val" else (np.int64 if k == "tile_off" else np.int32))
                           for k, v in tiles.items() if k.startswith("tile_")})
            self.n_items = self.n_seg_items = 0
            self.sell_entries, self.sell_seg_entries = tiles["entries"], tiles["seg_entries"]
            self.n_tiles = tiles["n_t4"] + tiles["n_t1"] + tiles["n_tseg"] + tiles["n_tfin"]
            self.tile_groups = int(ipw)
            tile_counts = (int(ipw), tiles["n_t4"], tiles["n_t1"], tiles["n_tseg"], tiles["n_tfin"], tiles["kmax"])
        else:
            sell = self._sell64(rowptr, col, val, deg, seg_beg[so], seg_len[so], so, seg_row[so], short_rows[ro])
            self.t.update({k: t(v, np.float32 if k == "val" else np.int32) for k, v in sell.items() if not k.startswith("_")})
            self.n_items, self.n_seg_items = sell["_n_items"], sell["_n_seg_items"]
            self.sell_entries, self.sell_seg_entries = sell["_entries"], sell["_seg_entries"]
            self.n_tiles, self.tile_groups = 0, 0
            tile_counts = (0, 0, 0, 0, 0, 0)
        self.desc = _lib.SellDesc(self.n_rows, self.n_src, self.n_items, self.n_seg_items, self.n_seg, self.n_long,
                                  p("item_dst"), p("item_len"), p("blk_off"), p("col"), p("val"), p("long_rows"),
                                  p("long_seg_ptr"), p("long_index"), p("rowptr"), p("csr_col"), p("csr_val"), p("item_long"),
                                  1 if tiered else 0, self.n_w1, self.n_w4, *tile_counts,
                                  p("tile_off"), p("tile_len"), p("tile_dst"), p("tile_long"), p("tile_col"), p("tile_val"))
        self._partials = {}

    @classmethod
    def on_device(cls, rowptr, col, val, n_src, threshold=64, side_split=None, ipw=8, rows_from=0, share=None):
        """The tiered plan of a CSR that is ALREADY on the device (rowptr int64 [n + 1], col int32, val fp32: csrc/adj.hip's output, or
        a rank's part of an edge list too large to plan on a host), built there by csrc/plan.hip: the arrays of SellPlan(m,
        tiered=True, ...) bit for bit (tests/test_shard_gpu.py::test_device_plan_build_equals_the_host_plan), with two read-backs of a
        handful of counts. T1 / T2 / segment length follow the same rules and environment switches as the host form."""
        lib = _lib.load()
        dev = rowptr.device
        n_rows = int(rowptr.numel()) - 1
        T = int(threshold)
        T1 = (64 if ipw <= 8 else 32) * ipw
        T2 = 256 * ipw
        TS = T
        G = int(ipw)
        split = -1 if side_split is None else int(side_split)
        i32 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int32, device=dev)
        i64 = lambda n: torch.empty(max(int(n), 1), dtype=torch.int64, device=dev)
        order, long_rows, long_index, seg_ptr, counts = i32(n_rows), i32(n_rows + 1), i32(n_rows), i32(n_rows + 1), i64(8)
        ws = torch.empty(int(lib.elimrec_plan_workspace(n_rows, 1, 1)), dtype=torch.uint8, device=dev)
        _lib.check(lib.elimrec_plan_rows(_dev(rowptr, "rowptr", torch.int64), n_rows, T, T1, T2, TS, G, split, int(rows_from),
                                         _dev(order, "order", torch.int32), _dev(long_rows, "long_rows", torch.int32),
                                         _dev(long_index, "long_index", torch.int32), _dev(seg_ptr, "long_seg_ptr", torch.int32),
                                         _dev(counts, "counts", torch.int64), _dev(ws, "workspace", torch.uint8), ws.numel(), _stream()), "plan_rows")
        n_w4, n_w1, n_split, n_short, _, n_long, n_seg = (int(x) for x in counts[:7].tolist())       # read-back 1
        n_tiles = int(lib.elimrec_plan_tile_count(n_w4, n_w1, n_seg, n_short, G))
        n_tseg = -(-n_seg // (4 * G)) * 4
        need = int(lib.elimrec_plan_workspace(n_rows, n_seg, n_tiles))
        if ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
        tile_off, tile_len, tile_dst = i64(n_tiles + 1), i32(n_tiles * G), i32(n_tiles * G)
        tile_long, gb, gs, totals = torch.zeros(max(n_tseg * G, 1), dtype=torch.int32, device=dev), i64(n_tiles * G), i32(n_tiles * G), i64(4)
        _lib.check(lib.elimrec_plan_tiles(_dev(rowptr, "rowptr", torch.int64), n_rows, T, T1, T2, TS, G, split, int(rows_from),
                                          _dev(order, "order", torch.int32), _dev(long_rows, "long_rows", torch.int32),
                                          _dev(seg_ptr, "long_seg_ptr", torch.int32), n_w4, n_w1, n_split, n_short, n_long, n_seg,
                                          _dev(tile_off, "tile_off", torch.int64), _dev(tile_len, "tile_len", torch.int32),
                                          _dev(tile_dst, "tile_dst", torch.int32), _dev(tile_long, "tile_long", torch.int32),
                                          _dev(gb, "gb", torch.int64), _dev(gs, "gs", torch.int32), _dev(totals, "totals", torch.int64),
                                          _dev(ws, "workspace", torch.uint8), ws.numel(), _stream()), "plan_tiles")
        total, seg_entries, kmax = (int(x) for x in totals[:3].tolist())                             # read-back 2
        tile_col = torch.zeros(total + 128, dtype=torch.int32, device=dev)
        tile_val = torch.zeros(total + 128, dtype=torch.float32, device=dev)
        _lib.check(lib.elimrec_plan_scatter(n_tiles, G, _dev(tile_off, "tile_off", torch.int64), _dev(gb, "gb", torch.int64),
                                            _dev(tile_len, "tile_len", torch.int32), _dev(gs, "gs", torch.int32), _dev(col, "col", torch.int32),
                                            _dev(val, "val"), _dev(tile_col, "tile_col", torch.int32), _dev(tile_val, "tile_val"), _stream()),
                   "plan_scatter")
        self = object.__new__(cls)
        self.device, self.sweep = dev, None
        self.n_rows, self.n_src, self.nnz = n_rows, int(n_src), int(rowptr[-1])
        self.n_seg, self.n_long, self.threshold = n_seg, n_long, T
        self.tiered, self.n_w1, self.n_w4 = True, n_w1, n_w4
        self.t = dict(long_rows=long_rows[:max(n_long, 1)].clone() if n_long else torch.zeros(1, dtype=torch.int32, device=dev),
                      long_seg_ptr=seg_ptr[:n_long + 1].clone(), long_index=long_index,
                      tile_off=tile_off, tile_len=tile_len, tile_dst=tile_dst, tile_long=tile_long, tile_col=tile_col, tile_val=tile_val)
        if share is not None:
            self.t.update({k: share.t[k] for k in ("rowptr", "csr_col", "csr_val")})
            self.shared = ("rowptr", "csr_col", "csr_val")
        else:
            self.t.update(rowptr=rowptr, csr_col=col, csr_val=val)
            self.shared = ()
        self.n_items = self.n_seg_items = 0
        self.sell_entries, self.sell_seg_entries = total, seg_entries
        self.n_tiles, self.tile_groups = n_tiles, G
        n_t1 = -(-n_w1 // 4) * 4
        n_tfin = n_tiles - 4 * n_w4 - n_t1 - n_tseg
        p = lambda k: self.t[k].data_ptr() if k in self.t else None
        self.desc = _lib.SellDesc(self.n_rows, self.n_src, 0, 0, self.n_seg, self.n_long, None, None, None, None, None, p("long_rows"),
                                  p("long_seg_ptr"), p("long_index"), p("rowptr"), p("csr_col"), p("csr_val"), None,
                                  1, self.n_w1, self.n_w4, G, 4 * n_w4, n_t1, n_tseg, n_tfin, max(kmax, 1),
                                  p("tile_off"), p("tile_len"), p("tile_dst"), p("tile_long"), p("tile_col"), p("tile_val"))
        self._partials = {}
        return self

    @staticmethod
    def _sell64(rowptr, col, val, deg, seg_beg, seg_len, seg_slot, seg_long, fin_rows):
        """SELL-64 work items (two-launch form): segment items, then the unsplit rows, super blocks of 64, (col, val)
        transposed inside a super block."""
        n_seg = len(seg_beg)
        pad = lambda n: (-n) % 64
        n_seg_items = n_seg + pad(n_seg)
        n_fin = len(fin_rows)
        n_items = n_seg_items + n_fin + pad(n_fin)
        item_dst = np.full(n_items, -1, np.int32)
        item_len = np.zeros(n_items, np.int32)
        item_beg = np.zeros(n_items, np.int64)
        item_dst[:n_seg] = seg_slot.astype(np.int32)
        item_len[:n_seg] = seg_len
        item_beg[:n_seg] = seg_beg
        item_dst[n_seg_items:n_seg_items + n_fin] = fin_rows.astype(np.int32)
        item_len[n_seg_items:n_seg_items + n_fin] = deg[fin_rows]
        item_beg[n_seg_items:n_seg_items + n_fin] = rowptr[fin_rows]
        nb = n_items // 64
        blk_len = item_len.reshape(nb, 64).max(1).astype(np.int64) if nb else np.zeros(0, np.int64)
        blk_off = np.concatenate([[0], np.cumsum(blk_len)]).astype(np.int64)
        total = int(blk_off[-1]) * 64
        if total >= 2 ** 31:
            raise ValueError("graph too large for int32 SELL offsets")
        sell_col = np.zeros(max(total, 1), np.int32)
        sell_val = np.zeros(max(total, 1), np.float32)
        # scatter every non-zero of every item to (blk_off[b] + j) * 64 + i
        it = np.repeat(np.arange(n_items), item_len)
        j = np.arange(len(it)) - np.repeat(np.cumsum(item_len) - item_len, item_len)
        src = item_beg[it] + j
        dstpos = (blk_off[it // 64] + j) * 64 + (it % 64)
        sell_col[dstpos] = col[src]
        sell_val[dstpos] = val[src]
        item_long = np.zeros(max(n_seg_items, 1), np.int32)
        item_long[:n_seg] = seg_long
        return dict(item_dst=item_dst, item_len=item_len, blk_off=blk_off, col=sell_col, val=sell_val, item_long=item_long,
                    _n_items=int(n_items), _n_seg_items=int(n_seg_items), _entries=total,
                    _seg_entries=int(blk_off[n_seg_items // 64]) * 64)

    @staticmethod
    def _wave_tiles(G, rowptr, col, val, deg, w4, w1, seg_beg, seg_len, seg_slot, seg_long, fin_rows):
        """Wave tiles of the one-launch form (struct elimrec_sell, tiered plan): per tile and lane group the first CSR
        position, the count and the stride of its neighbours, then one scatter of every non-zero to
        tile_off[tile] + step * G + group."""
        i64 = np.int64
        g = np.arange(G, dtype=i64)[None, :]
        def row_tiles(beg, n, dst):          # a contiguous run of a row per tile, dealt round-robin to the groups
            beg, n = np.asarray(beg, i64)[:, None], np.asarray(n, i64)[:, None]
            return beg + g, np.maximum(0, (n - g + G - 1) // G), np.full((len(beg), G), G, i64), np.repeat(np.asarray(dst, i64)[:, None], G, 1)
        def item_tiles(beg, n, dst):         # G items per tile, a group each
            k = len(beg)
            k_pad = (-k) % (4 * G)
            z = lambda a, fill: np.concatenate([np.asarray(a, i64), np.full(k_pad, fill, i64)]).reshape(-1, G)
            return z(beg, 0), z(n, 0), np.ones(((k + k_pad) // G, G), i64), z(dst, -1)
        def pad4(parts):                     # empty tiles up to a multiple of 4
            k = (-len(parts[0])) % 4
            if k == 0:
                return parts
            e = (np.zeros((k, G), i64), np.zeros((k, G), i64), np.ones((k, G), i64), np.full((k, G), -1, i64))
            return tuple(np.concatenate([a, b]) for a, b in zip(parts, e))
        n4 = deg[w4].astype(i64)
        q = (((n4 + 3) // 4 + G - 1) // G) * G                          # quarter length, a multiple of G
        cidx = np.arange(4, dtype=i64)[None, :]
        cb = (rowptr[w4].astype(i64)[:, None] + cidx * q[:, None]).reshape(-1)
        cn = np.clip(n4[:, None] - cidx * q[:, None], 0, q[:, None]).reshape(-1)
        A = row_tiles(cb, cn, np.repeat(w4, 4))
        B = pad4(row_tiles(rowptr[w1], deg[w1], w1))
        C = item_tiles(seg_beg, seg_len, seg_slot)
        D = item_tiles(rowptr[fin_rows], deg[fin_rows], fin_rows)
        GB, GL, GS, GD = (np.concatenate([a, b, c, d]) for a, b, c, d in zip(A, B, C, D))
        steps = GL.max(1) if len(GL) else np.zeros(0, i64)
        tile_off = np.concatenate([[0], np.cumsum(steps * G)]).astype(i64)
        total = int(tile_off[-1])          # may exceed 2^31: tile_off is int64 on the device too
        tcol = np.zeros(total + 128, np.int32)                         # a wave reads whole 64-entry lines past its tile
        tval = np.zeros(total + 128, np.float32)
        glf, gbf, gsf = GL.reshape(-1), GB.reshape(-1), GS.reshape(-1)
        it = np.repeat(np.arange(len(glf), dtype=i64), glf)
        j = np.arange(len(it), dtype=i64) - np.repeat(np.cumsum(glf) - glf, glf)
        src = gbf[it] + j * gsf[it]
        pos = tile_off[it // G] + j * G + (it % G)
        tcol[pos] = col[src]
        tval[pos] = val[src]
        n_tseg = len(C[0])
        tile_long = np.zeros(max(n_tseg * G, 1), np.int32)
        tile_long[:len(seg_long)] = seg_long
        seg_end = len(A[0]) + len(B[0]) + n_tseg
        return dict(tile_off=tile_off, tile_len=GL.reshape(-1), tile_dst=GD.reshape(-1), tile_long=tile_long, tile_col=tcol,
                    tile_val=tval, n_t4=len(A[0]), n_t1=len(B[0]), n_tseg=n_tseg, n_tfin=len(D[0]), entries=total,
                    seg_entries=int(tile_off[seg_end]), kmax=int(((steps * G + 63) // 64).max()) if len(steps) else 1)

    def ref(self):
        return ctypes.byref(self.desc)

    def partials(self, ns, w):
        """Scratch for the split rows' partial sums, one buffer per table geometry."""
        key = (ns, w)
        if key not in self._partials:
            nbytes = int(_lib.load().elimrec_slab_partials_bytes(self.ref(), ns, w))
            self._partials[key] = torch.zeros(max(nbytes // 4, 1), dtype=torch.float32, device=self.device)   # + arrival counters
        return self._partials[key]

    def index_bytes(self):
        """Bytes of index data one pass of a hop reads (SELL col + val, item records, block offsets)."""
        if self.tiered:
            return 8 * self.sell_entries + (8 * self.tile_groups + 8) * self.n_tiles
        return 8 * self.sell_entries + 8 * self.n_items + 4 * (self.n_items // 64 + 1)


class SweepPlan(object):
    """The rows [0, n_sweep) of a bipartite adjacency -- one side, whose sources [n_sweep, n) are a table far beyond the caches --
    as the window-sweep hop takes them (csrc/sweep.hip, elimrec_slab_sweep_hop), beside a tile plan of the OTHER side's rows:
    a full hop of `plan` is then the tile hop over `items` (rows >= n_sweep, which gather from the small side) + the sweep over
    the rows below (models/EliMRec.py:243-247; BASELINE.json configs[3] / [4]: 36 k / 1 M user rows gathering from 1.2 M / 100 M
    item rows). Row blocks and entry streams depend on the table geometry (LDS rows per workgroup, XCD roles per slab) and are
    made per (ns, w)."""

    BPX = 32                                                          # workgroups per XCD role (32 CUs per XCD)

    def __init__(self, plan, m, n_sweep, device, threshold, ipw):
        m = m.tocsr()
        m.sort_indices()
        self.n_sweep, self.n = int(n_sweep), int(m.shape[0])
        nnz = int(m.indptr[self.n_sweep])
        assert m.shape[0] == m.shape[1] and (nnz == 0 or int(m.indices[:nnz].min()) >= self.n_sweep), \
            "the swept rows must gather from rows >= n_sweep only"
        self.items = SellPlan(m, device, threshold=threshold, side_split=None, tiered=True, ipw=ipw, rows_from=self.n_sweep, share=plan)
        self.row_nnz = np.diff(m.indptr[:self.n_sweep + 1]).astype(np.int64)
        self._col, self._val = m.indices[:nnz].astype(np.int32), m.data[:nnz].astype(np.float32)
        self.device = torch.device(device)
        self._geo = {}

    def window(self, w):
        """Source rows per window: 2 MB of row pieces, half an XCD's L2 (measured at the configs[3] shape, 128-B pieces, us per hop /
        GB past L2 of the sweep launch: 8 192 rows 927 / 1.64, 12 288 892, 16 384 879, 24 576 875 / 2.26, 32 768 875 -- the time is flat
        from 16 384 on, the traffic is not)."""
        return int(os.environ.get("ELIMREC_SWEEP_WINDOW", (2 << 20) // (4 * w)))

    def geometry(self, ns, w):
        """For tables of ns slabs x w floats: the swept rows cut into parts x passes x bpx contiguous blocks of about equal
        non-zero counts (none above the LDS capacity), their entries ordered by (block, source window, row, column) -- a stable
        sort of the CSR order -- and every (block, window) cut at row boundaries into 64 chunks of about equal length."""
        key = (ns, w)
        if key not in self._geo:
            cap = int(_lib.load().elimrec_slab_sweep_lds_rows(int(w)))
            parts, bpx, n = 8 // min(ns, 8), self.BPX, self.n_sweep
            passes = max(1, -(-n // (parts * bpx * cap)))
            nb = parts * passes * bpx
            cum = np.concatenate([[0], np.cumsum(self.row_nnz)])
            want = np.searchsorted(cum, np.arange(1, nb) * (cum[-1] / nb))          # equal non-zeros ...
            ptr = np.zeros(nb + 1, np.int64)
            ptr[-1] = n
            for k in range(1, nb):                                                   # ... under the row cap, leaving room for the rest
                ptr[k] = min(max(int(want[k - 1]), ptr[k - 1], n - (nb - k) * cap), ptr[k - 1] + cap, n)
            assert np.all(np.diff(ptr) >= 0) and np.diff(ptr).max() <= cap, "sweep blocks do not fit"
            rows = np.repeat(np.arange(n, dtype=np.int64), self.row_nnz)
            blk = np.searchsorted(ptr, rows, side="right") - 1
            rloc = rows - ptr[blk]
            win = (self._col.astype(np.int64) - n) // self.window(w)
            n_win = int(win.max()) + 1 if len(win) else 1
            # rows to the waves of their block's workgroup: by decreasing length, dealt forth and back -- equal non-zero counts
            lpr = w // 4
            G, NW = 64 // lpr, lpr
            blk_of_row = np.searchsorted(ptr, np.arange(n), side="right") - 1
            by_len = np.lexsort((-self.row_nnz, blk_of_row))
            pos = np.arange(n) - ptr[blk_of_row[by_len]]
            snake = np.where((pos // NW) % 2 == 0, pos % NW, NW - 1 - pos % NW)
            wave_of_row = np.empty(n, np.int64)
            wave_of_row[by_len] = snake
            # entries by (block, wave, window), rows ascending inside (stable: the CSR order is kept), then every such cell cut at
            # row boundaries into G chunks of about equal length
            cell = (blk * NW + wave_of_row[rows]) * n_win + win
            order = np.argsort(cell, kind="stable")
            n_cells = nb * NW * n_win
            cell_ptr = np.concatenate([[0], np.cumsum(np.bincount(cell, minlength=n_cells))]).astype(np.int64)
            srow, scell = rows[order], cell[order]
            first = np.concatenate([[True], (srow[1:] != srow[:-1]) | (scell[1:] != scell[:-1])]) if len(srow) else np.zeros(0, bool)
            starts = np.nonzero(first)[0]                                            # cut points allowed: an entry that opens a row of a cell
            k = np.arange(G + 1, dtype=np.int64)[None, :]
            ideal = (cell_ptr[:-1, None] + ((cell_ptr[1:] - cell_ptr[:-1])[:, None] * k) // G).reshape(-1)
            cut = starts[np.minimum(np.searchsorted(starts, ideal), len(starts) - 1)] if len(starts) else ideal
            cut = np.minimum(np.maximum(cut.reshape(-1, G + 1), cell_ptr[:-1, None]), cell_ptr[1:, None])
            cut[:, 0], cut[:, -1] = cell_ptr[:-1], cell_ptr[1:]
            cut = np.maximum.accumulate(cut, axis=1)
            # every chunk as 80-byte step records (8 source rows | 8 values | 8 block-relative rows, 16 bit, 0xffff = none); all G
            # chunks of a (block, wave, window) cell get the cell's longest chunk's record count, and the records go [step][group]
            csz = cut[:, 1:] - cut[:, :-1]                                         # [cells, G] entries per chunk
            cell_steps = ((csz + 7) // 8).max(axis=1)                              # steps of a cell
            step_ptr = np.concatenate([[0], np.cumsum(cell_steps)]).astype(np.int64)
            n_steps = int(step_ptr[-1])
            dummy = int(np.diff(ptr).max())                                        # the row behind every block's rows
            rec = np.zeros(((n_steps + 8) * G, 20), np.int32)                      # (+ the kernel's read-ahead past the last step)
            rows16 = np.full(((n_steps + 8) * G, 8), dummy, np.uint16)
            chunk_of = np.repeat(np.arange(csz.size, dtype=np.int64), csz.reshape(-1))   # chunk of every (sorted) entry
            p_in = np.arange(len(chunk_of), dtype=np.int64) - np.repeat(cut[:, :-1].reshape(-1), csz.reshape(-1))
            ri = (step_ptr[chunk_of // G] + p_in // 8) * G + chunk_of % G
            si = p_in % 8
            rec[ri, si] = (self._col[order].astype(np.int64) * lpr).astype(np.uint32).view(np.int32)
            rec[ri, 8 + si] = self._val[order].view(np.int32)
            rows16[ri, si] = rloc[order].astype(np.uint16)
            rec[:, 16:20] = rows16.view(np.int32)
            slot_ptr = step_ptr[::n_win]                                           # per (block, wave): its cells are consecutive
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(self.device)
            self._geo[key] = dict(block_ptr=t(ptr, np.int32), slot_ptr=t(slot_ptr, np.int64), rec=t(rec.reshape(-1), np.int32),
                                  parts=parts, passes=passes, bpx=bpx, rows=int(np.diff(ptr).max()), n_win=n_win, n_rec=n_steps * G,
                                  fill=float(len(order)) / max(8 * n_steps * G, 1))
        return self._geo[key]

    def index_bytes(self, ns, w):
        """Index bytes one hop reads: the records of the swept side + the tile plan of the other."""
        return 80 * int(self.geometry(ns, w)["n_rec"]) + self.items.index_bytes()

    def device_bytes(self):
        """Device memory of this plan beyond the whole-graph plan it sits beside."""
        own = sum(v.numel() * v.element_size() for k, v in self.items.t.items() if k not in self.items.shared)
        own += sum(v.numel() * v.element_size() for v in self.items._partials.values())
        return own + sum(v.numel() * v.element_size() for g in self._geo.values() for v in g.values() if hasattr(v, "numel"))

    def hop(self, xin, xout, add=None, add_mask=None, scale=1.0):
        ns, w = xin.ns, xin.w
        g = self.geometry(ns, w)
        _lib.check(_lib.load().elimrec_slab_sweep_hop(_dev(g["slot_ptr"], "slot_ptr", torch.int64), _dev(g["rec"], "records", torch.int32), self.n, self.n,
                                                      _dev(g["block_ptr"], "block_ptr", torch.int32), g["parts"], g["passes"], g["bpx"],
                                                      g["rows"], ns, w, _dev(xin.data, "xin"), _dev(xout.data, "xout"),
                                                      _dev(None if add is None else add.data, "add"),
                                                      _dev(add_mask, "add_mask", torch.int32), float(scale), _stream()), "slab_sweep_hop")

    def hop_adam(self, xin, grad_out, add, add_mask, scale, p_in, p_out, m, v, lr, beta1, beta2, eps, weight_decay, step):
        """elimrec_slab_sweep_hop_adam: the swept rows of the adjoint's last hop, their Adam step as the launch's epilogue."""
        ns, w = xin.ns, xin.w
        g = self.geometry(ns, w)
        _lib.check(_lib.load().elimrec_slab_sweep_hop_adam(
            _dev(g["slot_ptr"], "slot_ptr", torch.int64), _dev(g["rec"], "records", torch.int32), self.n, self.n,
            _dev(g["block_ptr"], "block_ptr", torch.int32), g["parts"], g["passes"], g["bpx"], g["rows"], ns, w, _dev(xin.data, "xin"),
            _dev(None if grad_out is None else grad_out.data, "grad"), _dev(None if add is None else add.data, "add"),
            _dev(add_mask, "add_mask", torch.int32), float(scale), _dev(p_in, "p_in"), _dev(p_out, "p_out"), _dev(m, "m"), _dev(v, "v"),
            float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step), _stream()), "slab_sweep_hop_adam")


SWEEP_AUTO_MAX_NNZ = 1 << 27


def sweep_wanted(n_rows, dl, nnz=None):
    """Tables whose column slice is beyond the Infinity Cache (256 MB) take the window-sweep form for the side that gathers from
    the large side (ELIMREC_SWEEP=1 / 0 forces it on / off). Automatically only up to 2^27 non-zeros: the sweep plan costs 23 B
    per non-zero on top of the tile plan and its geometry is built by numpy on the host -- at BASELINE.json configs[4]'s 2e9
    non-zeros that is 43 GiB per GPU and hours, so that shape keeps the tile hop unless asked (capacity.plan counts either)."""
    env = os.environ.get("ELIMREC_SWEEP", "")
    if env in ("0", "1"):
        return env == "1"
    return n_rows * dl * 4 > (256 << 20) and (nnz is None or nnz <= SWEEP_AUTO_MAX_NNZ)


def sweep_tiles_xcds(ns):
    """The window sweep gives every XCD one slab role (elimrec_slab_sweep_hop): the slab count must divide, or be a multiple of,
    the 8 XCDs. A slice of 96, 48 or 192 columns (3 or 6 slabs) keeps the tile hop."""
    return (8 % ns == 0) if ns <= 8 else (ns % 8 == 0)


class SlabTable(object):
    """[n x (ns*w)] fp32 table stored slab-major in one flat tensor."""

    def __init__(self, n, ns, w, device, data=None):
        self.n, self.ns, self.w = int(n), int(ns), int(w)
        self.data = torch.empty(self.ns * self.n * self.w, dtype=torch.float32, device=device) if data is None else data
        assert self.data.numel() == self.ns * self.n * self.w and self.data.is_contiguous() and self.data.dtype == torch.float32

    @property
    def cols(self):
        return self.ns * self.w

    def like(self):
        return SlabTable(self.n, self.ns, self.w, self.data.device)

    def from_rows(self, src, col0=0):
        """Columns [col0, col0 + ns*w) of a row-major 2-D fp32 tensor (unit column stride)."""
        assert src.dim() == 2 and src.stride(1) == 1 and src.shape[0] == self.n and col0 + self.cols <= src.shape[1]
        _lib.check(_lib.load().elimrec_slab_from_rows(_dev(src, "src"), src.stride(0), int(col0), self.n, self.ns, self.w,
                                                      _dev(self.data, "slab"), _stream()), "slab_from_rows")
        return self

    def to_rows(self, dst, col0=0):
        assert dst.dim() == 2 and dst.stride(1) == 1 and dst.shape[0] == self.n and col0 + self.cols <= dst.shape[1]
        _lib.check(_lib.load().elimrec_slab_to_rows(_dev(self.data, "slab"), self.n, self.ns, self.w, _dev(dst, "dst"),
                                                    dst.stride(0), int(col0), _stream()), "slab_to_rows")
        return dst

    def dense(self):
        """Row-major fp32 copy [n x cols] (tests, checkpoints)."""
        return self.to_rows(torch.empty(self.n, self.cols, dtype=torch.float32, device=self.data.device))


def source_bits(plan, ns, w, gs, src_mask):
    """elimrec_slab_source_bits: the masked hop's per-index-line source bits for this bitmap, ahead of the hop."""
    part = plan.partials(ns, w)
    _lib.check(_lib.load().elimrec_slab_source_bits(plan.ref(), ns, w, int(gs), _dev(src_mask, "src_mask", torch.int32),
                                                    _dev(part, "partials"), part.numel() * 4, _stream()), "slab_source_bits")


def _swept(plan, xin):
    """Whether whole hops of this plan on tables of xin's geometry take the two-launch form with a swept side."""
    return plan.sweep is not None and xin.w in (16, 32) and sweep_tiles_xcds(xin.ns)


def hop(plan, xin, xout, gs=None, src_mask=None, add=None, add_mask=None, scale=1.0, seg_only=False, bits_ready=False, bwd_w=None, bwd_w_phase=1):
    """xout = (A . xin + [add_mask] add) * scale on slab tables of equal geometry. seg_only: xout is a flat fp32 tensor
    [ns x n_long x w] receiving the split rows only.
    bwd_w: the handle of ops.linear_bwd_w_batched(..., defer_reduce=True / defer_all=True) -- bwd_w_phase 1: its slab reduce,
    0: its partial launch, run as extra workgroups of this launch (fp32 tables, tiered plan: elimrec_slab_hop_bwd_w)."""
    ns, w = xin.ns, xin.w
    gs = choose_groups(ns) if gs is None else gs
    if _swept(plan, xin) and src_mask is None and not seg_only and isinstance(xout, SlabTable):
        # a whole hop of a graph with a swept side: the tile hop over the other side's rows (it carries the weight gradients' extra
        # workgroups, if any), the window sweep over this side's
        hop(plan.sweep.items, xin, xout, gs=gs, add=add, add_mask=add_mask, scale=scale, bwd_w=bwd_w, bwd_w_phase=bwd_w_phase)
        plan.sweep.hop(xin, xout, add=add, add_mask=add_mask, scale=scale)
        return
    part = plan.partials(ns, w)
    out = xout if isinstance(xout, torch.Tensor) else xout.data
    lib = _lib.load()
    if bwd_w is not None:
        arr, n, wsp = bwd_w
        _lib.check(lib.elimrec_slab_hop_bwd_w(plan.ref(), ns, w, int(gs), _dev(xin.data, "xin"), _dev(src_mask, "src_mask", torch.int32),
                                               _dev(out, "xout"), _dev(None if add is None else add.data, "add"),
                                               _dev(add_mask, "add_mask", torch.int32), float(scale), _dev(part, "partials"),
                                               part.numel() * 4, 2 if bits_ready else 0, arr, n, _dev(wsp, "workspace", torch.uint8),
                                               wsp.numel(), int(bwd_w_phase), _stream()), "slab_hop_bwd_w")
        return
    _lib.check(lib.elimrec_slab_hop(plan.ref(), ns, w, int(gs), _dev(xin.data, "xin"),
                                    _dev(src_mask, "src_mask", torch.int32), _dev(out, "xout"),
                                    _dev(None if add is None else add.data, "add"),
                                    _dev(add_mask, "add_mask", torch.int32), float(scale), _dev(part, "partials"),
                                    part.numel() * 4, (1 if seg_only else 0) | (2 if bits_ready else 0), _stream()), "slab_hop")


def hop_adam(plan, xin, grad_out, gs, add, add_mask, scale, p_in, p_out, m, v, lr, beta1, beta2, eps, weight_decay, step, tail_jobs=(),
             loss_sum=None):
    """elimrec_slab_hop_adam: the hop whose output is the gradient of the fp32 slab table p_in, consumed in place by an
    Adam step (p_out / m / v flat fp32 of the table's geometry). grad_out: a table to also receive the gradient, or None.
    tail_jobs: _lib.AdamJob spans (the projection weights) updated by extra workgroups of the same launch.
    loss_sum = (loss_rows, loss_out): one more workgroup adds the loss rows in elimrec_sum's order into loss_out."""
    if isinstance(tail_jobs, tuple) and len(tail_jobs) == 2 and not isinstance(tail_jobs[0], _lib.AdamJob):
        arr, n_tail = tail_jobs                      # (persistent AdamJob array, count): the caller keeps it alive and in place
    else:
        arr, n_tail = ((_lib.AdamJob * len(tail_jobs))(*tail_jobs) if tail_jobs else None), len(tail_jobs)
    ns, w = xin.ns, xin.w
    if _swept(plan, xin):
        # the other side's rows by the tile hop (with the optimizer spans and the loss sum as its extra workgroups), the swept side's
        # by the window sweep: both with the Adam step as their epilogue, on disjoint rows of the same buffers
        hop_adam(plan.sweep.items, xin, grad_out, gs, add, add_mask, scale, p_in, p_out, m, v, lr, beta1, beta2, eps, weight_decay, step,
                 tail_jobs=(arr, n_tail), loss_sum=loss_sum)
        plan.sweep.hop_adam(xin, grad_out, add, add_mask, scale, p_in, p_out, m, v, lr, beta1, beta2, eps, weight_decay, step)
        return
    part = plan.partials(ns, w)
    _lib.check(_lib.load().elimrec_slab_hop_adam(
        plan.ref(), ns, w, int(gs), _dev(xin.data, "xin"), _dev(None if grad_out is None else grad_out.data, "grad"),
        _dev(None if add is None else add.data, "add"), _dev(add_mask, "add_mask", torch.int32), float(scale), _dev(part, "partials"),
        part.numel() * 4, _dev(p_in, "p_in"), _dev(p_out, "p_out"), _dev(m, "m"), _dev(v, "v"), float(lr), float(beta1), float(beta2),
        float(eps), float(weight_decay), int(step), arr, n_tail, _dev(None if loss_sum is None else loss_sum[0], "loss_rows"),
        0 if loss_sum is None else loss_sum[0].numel(), _dev(None if loss_sum is None else loss_sum[1], "loss_out"), _stream()),
        "slab_hop_adam")



def rows(plan, ns, w, L, U, layers, long_tab, row_ids, counts, R, n_lists, out0, narrow, narrow_by_node):
    """Layer means at listed rows (elimrec_slab_rows). layers: list of L+1 flat tensors (last may be None)."""
    ptrs = (ctypes.c_void_p * (L + 1))(*[None if t is None else _dev(t, "layer") for t in layers])
    assert out0.stride(1) == 1 and narrow.stride(1) == 1
    _lib.check(_lib.load().elimrec_slab_rows(plan.ref(), ns, w, L, int(U), ptrs, _dev(long_tab, "long_tab"),
                                             _dev(row_ids, "rows", torch.int32), _dev(counts, "counts", torch.int32),
                                             int(R), int(n_lists), _dev(out0, "out0"), out0.stride(0),
                                             _dev(narrow, "narrow"), narrow.stride(0), 1 if narrow_by_node else 0,
                                             _stream()), "slab_rows")


def merge_rows(all_rows, all_keys, world, U, I, srcA, srcB, mask, M=0):
    """[H | G] rows (M = 0) or dOut rows of M column blocks (M >= 1) of `world` ranks -> slab-major adjoint sources + row
    bitmap (elimrec_slab_merge_rows). M = -1: [H | G] rows, H always into srcA and G always into srcB (the wide form)."""
    R = all_keys.numel() // world
    assert all_rows.is_contiguous() and all_rows.shape == (world * R, (M if M > 0 else 2) * srcA.cols) and mask.numel() * 32 >= U + I
    _lib.check(_lib.load().elimrec_slab_merge_rows(_dev(all_rows, "rows"), _dev(all_keys, "keys", torch.int32), int(world), R,
                                                   int(U), int(I), srcA.ns, srcA.w, int(M), _dev(srcA.data, "srcA"),
                                                   _dev(srcB.data, "srcB"), _dev(mask, "mask", torch.int32), _stream()),
               "slab_merge_rows")


def adam_step_out(p_in, p_out, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    n = p_in.numel()
    for t in (p_in, p_out, g, m, v):
        assert t.is_contiguous() and t.numel() == n
    _lib.check(_lib.load().elimrec_adam_step_out(_dev(p_in, "p_in"), _dev(p_out, "p_out"), _dev(g, "g"), _dev(m, "m"),
                                                 _dev(v, "v"), n, float(lr), float(beta1), float(beta2), float(eps),
                                                 float(weight_decay), int(step), _stream()), "adam_step_out")


def rows_bitmap(keys, N, mask):
    """mask <- row bitmap of the key lists keys [W x R] (elimrec_rows_bitmap): the words merge_rows writes for the same lists."""
    W, R = keys.shape
    assert keys.is_contiguous() and mask.numel() * 32 >= N
    _lib.check(_lib.load().elimrec_rows_bitmap(_dev(keys, "keys", torch.int32), int(W), int(R), int(N), _dev(mask, "mask", torch.int32),
                                               _stream()), "rows_bitmap")


def wide_from_master(master, U, wide):
    """Layer 0 of the wide form (csrc/wide.hip): wide [N x 2 dl] <- [E_u | 0] on user rows, [0 | E_i] on item rows."""
    assert wide.ns == 2 * master.ns and wide.w == master.w and wide.n == master.n
    _lib.check(_lib.load().elimrec_wide_from_master(_dev(master.data, "master"), int(U), master.n, master.ns, master.w, _dev(wide.data, "wide"),
                                                    _stream()), "wide_from_master")


def wide_rows(layers, n, ns, w, rows, total, out0, narrow):
    """(layer mean, shared part) of listed rows from the L+1 wide layer tables (flat tensors): out0 = mean_k (left + right),
    narrow = mean_k left. rows: int32 ids (negative = padding) or None for rows 0 .. total-1."""
    L = len(layers) - 1
    ptrs = (ctypes.c_void_p * (L + 1))(*[_dev(t, "layer") for t in layers])
    assert out0.stride(1) == 1 and narrow.stride(1) == 1
    _lib.check(_lib.load().elimrec_wide_rows(ptrs, L, int(n), int(ns), int(w), _dev(rows, "rows", torch.int32), int(total), _dev(out0, "out0"),
                                             out0.stride(0), _dev(narrow, "narrow"), narrow.stride(0), _stream()), "wide_rows")


def wide_grad(wide, U, scale, grad):
    """grad [N x dl] <- scale * (left half of the wide adjoint table on user rows, right half on item rows)."""
    assert wide.ns == 2 * grad.ns and wide.w == grad.w and wide.n == grad.n
    _lib.check(_lib.load().elimrec_wide_grad(_dev(wide.data, "wide"), int(U), grad.n, grad.ns, grad.w, float(scale), _dev(grad.data, "grad"),
                                             _stream()), "wide_grad")
