"""Device-resident helpers: torch owns the HBM buffers and the stream, the C ABI does the work.

torch is plumbing only (allocation, streams, torch.distributed in bench.py); every
function here ends in a call through ``include/avrecode_ms_amd.h``.
"""
from __future__ import annotations

import ctypes

from . import KIND_CABAC, KIND_RANGE, NOP_CABAC, NOP_RANGE, AvrError, SynthConfig, _check, lib


def synth_config(workload: int, scale_permille: int = 1000, first_slice: int = 0) -> SynthConfig:
    cfg = SynthConfig()
    _check(lib().avr_synth_config_init(ctypes.byref(cfg), workload, scale_permille, first_slice))
    return cfg


def _stream_ptr(torch):
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def plan_tiles(n_bins):
    """Processing order and tile offsets for per-slice bin counts (int32 tensor, any device).

    Slices are processed longest first, 64 per tile (one wave); a tile is as long as its
    longest slice.  Returns (order int32[n], tile_off int64[n_tiles+1] in 16-byte chunks).
    Same plan as plan_tiles() in csrc/avr_api.cpp.
    """
    import torch
    n = n_bins.numel()
    order = torch.argsort(n_bins.to(torch.int64), descending=True, stable=True)
    chunks = (n_bins.to(torch.int64)[order] + 7) // 8
    tile_max = chunks[::64]
    tile_off = torch.zeros(tile_max.numel() + 1, dtype=torch.int64, device=n_bins.device)
    tile_off[1:] = torch.cumsum(tile_max * 64, 0)
    return order.to(torch.int32), tile_off


def encode_tiles(kind, tiles, tile_off, n_bins, order, out, out_off, out_len, status,
                 init_states=None, n_states=0, final_states=None, device_index=0):
    """Enqueue K1 (kind 0) or K2 (kind 1) on torch's current stream."""
    import torch
    L = lib()
    n = n_bins.numel()
    sp = _stream_ptr(torch)
    if kind == KIND_CABAC:
        _check(L.avr_cabac_encode_tiles_device(
            device_index, sp, tiles.data_ptr(), tile_off.data_ptr(), n_bins.data_ptr(), order.data_ptr(), n,
            init_states.data_ptr() if init_states is not None else None, n_states,
            out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), status.data_ptr(),
            final_states.data_ptr() if final_states is not None else None))
    else:
        _check(L.avr_range_encode_tiles_device(
            device_index, sp, tiles.data_ptr(), tile_off.data_ptr(), n_bins.data_ptr(), order.data_ptr(), n,
            out.data_ptr(), out_off.data_ptr(), out_len.data_ptr(), status.data_ptr()))


class DeviceWorkload:
    """A batch of slices resident in HBM in the wave-interleaved tile layout."""

    def __init__(self, kind, device_index, n_bins, order, tile_off, tiles, init_states, n_states):
        import torch
        self.kind, self.device_index = kind, device_index
        self.n_bins, self.order, self.tile_off, self.tiles = n_bins, order, tile_off, tiles
        self.init_states, self.n_states = init_states, n_states
        self.n_slices = n_bins.numel()
        dev = n_bins.device
        cap = (n_bins.to(torch.int64) + 16 + 7) // 8 * 8          # worst case 8 bits per bin + stop bytes
        self.out_off = torch.zeros(self.n_slices + 1, dtype=torch.int64, device=dev)
        self.out_off[1:] = torch.cumsum(cap, 0)
        self.out = torch.empty(int(self.out_off[-1].item()), dtype=torch.uint8, device=dev)
        self.out_len = torch.zeros(self.n_slices, dtype=torch.int32, device=dev)
        self.status = torch.zeros(self.n_slices, dtype=torch.int32, device=dev)
        self.final_states = (torch.empty(self.n_slices * max(n_states, 1), dtype=torch.uint8, device=dev)
                             if kind == KIND_CABAC else None)
        self.total_bins = int(n_bins.to(torch.int64).sum().item())
        # K1 sized by the context count of the previous run of this object (avr_cabac_encode_*_device_hinted): no host round trip
        # in the call; settle() looks at what the device reported once the caller has synchronised
        self.rows_hint, self._hinted_path = 0, None
        self._counts = torch.zeros(2, dtype=torch.int32).pin_memory() if dev.type == "cuda" and kind == KIND_CABAC else None

    @classmethod
    def synth(cls, workload, n_slices, kind=KIND_CABAC, device_index=0, scale_permille=1000, first_slice=0):
        """Generate BASELINE.json config `workload` on the device (avr_synth_*_device)."""
        import torch
        L = lib()
        dev = torch.device("cuda", device_index)
        cfg = synth_config(workload, scale_permille, first_slice)
        with torch.cuda.device(dev):
            sp = _stream_ptr(torch)
            n_bins = torch.zeros(n_slices, dtype=torch.int32, device=dev)
            _check(L.avr_synth_count_device(device_index, sp, ctypes.byref(cfg), kind, n_slices, n_bins.data_ptr()))
            order, tile_off = plan_tiles(n_bins)
            tiles = torch.empty(int(tile_off[-1].item()) * 16, dtype=torch.uint8, device=dev)
            init_states = (torch.empty(n_slices * cfg.n_states, dtype=torch.uint8, device=dev)
                           if kind == KIND_CABAC else None)
            _check(L.avr_synth_generate_tiles_device(
                device_index, sp, ctypes.byref(cfg), kind, n_slices, order.data_ptr(), tile_off.data_ptr(),
                tiles.data_ptr(), init_states.data_ptr() if init_states is not None else None))
            torch.cuda.synchronize(dev)
        w = cls(kind, device_index, n_bins, order, tile_off, tiles, init_states, cfg.n_states if kind == KIND_CABAC else 0)
        w.cfg = cfg
        return w

    @classmethod
    def from_host(cls, kind, recs_list, init_states_list=None, device_index=0):
        """Upload per-slice uint16 record arrays and pack them into tiles on the device."""
        import numpy as np
        import torch
        L = lib()
        dev = torch.device("cuda", device_index)
        n = len(recs_list)
        nb = np.array([len(r) for r in recs_list], dtype=np.int32)
        padded = (nb.astype(np.int64) + 7) // 8 * 8
        rec_off = np.zeros(n + 1, dtype=np.int64)
        rec_off[1:] = np.cumsum(padded)
        flat = np.full(int(rec_off[-1]) + 8, NOP_CABAC if kind == KIND_CABAC else NOP_RANGE, dtype=np.uint16)
        for i, r in enumerate(recs_list):
            flat[rec_off[i]:rec_off[i] + len(r)] = r
        n_states = len(init_states_list[0]) if (kind == KIND_CABAC and init_states_list) else 0
        with torch.cuda.device(dev):
            d_flat = torch.from_numpy(flat.view(np.int16)).to(dev)
            d_off = torch.from_numpy(rec_off).to(dev)
            n_bins = torch.from_numpy(nb).to(dev)
            order, tile_off = plan_tiles(n_bins)
            tiles = torch.empty(max(int(tile_off[-1].item()), 1) * 16, dtype=torch.uint8, device=dev)
            pack_status = torch.zeros(n, dtype=torch.int32, device=dev)
            _check(L.avr_pack_tiles_device(device_index, _stream_ptr(torch), kind, n_states, d_flat.data_ptr(),
                                           d_off.data_ptr(), n_bins.data_ptr(), order.data_ptr(), n, tile_off.data_ptr(),
                                           tiles.data_ptr(), pack_status.data_ptr()))
            init = None
            if kind == KIND_CABAC:
                init = torch.from_numpy(np.concatenate([np.asarray(s, dtype=np.uint8) for s in init_states_list])
                                        if n_states else np.zeros(1, np.uint8)).to(dev)
            torch.cuda.synchronize(dev)
        w = cls(kind, device_index, n_bins, order, tile_off, tiles, init, n_states)
        w.rec_flat, w.rec_off = d_flat, d_off
        w.status.copy_(pack_status)
        return w

    def encode(self):
        """One pass of the hot path over the batch (enqueued on torch's current stream).  K1: sized by the context count the previous
        run of this object reported (none yet: the call asks the device and waits); exact whatever the guess.  settle() after a
        synchronisation takes the count over for the next run."""
        if self.kind == KIND_CABAC and self._counts is not None:
            import torch
            self._hinted_path = "tiles"
            _check(lib().avr_cabac_encode_tiles_device_hinted(
                self.device_index, _stream_ptr(torch), self.tiles.data_ptr(), self.tile_off.data_ptr(), self.n_bins.data_ptr(),
                self.order.data_ptr(), self.n_slices, self.init_states.data_ptr() if self.init_states is not None else None, self.n_states,
                self.out.data_ptr(), self.out_off.data_ptr(), self.out_len.data_ptr(), self.status.data_ptr(),
                self.final_states.data_ptr() if self.final_states is not None else None, self.rows_hint, self._counts.data_ptr()))
            return
        encode_tiles(self.kind, self.tiles, self.tile_off, self.n_bins, self.order, self.out, self.out_off,
                     self.out_len, self.status, self.init_states, self.n_states, self.final_states, self.device_index)

    def _part_array(self, rows_hint):
        """avr_chunked_part[n_parts]: the batch's arrays from each part's first slice on, the part's own plan, workspace and counts."""
        from . import ChunkedPart
        recs, rec_off = self._slice_major()
        ns = max(self.n_states, 1)
        arr = (ChunkedPart * self.n_parts)()
        for i, p in enumerate(self._parts):
            a = p["first"]
            arr[i] = ChunkedPart(
                rec_off.data_ptr() + 8 * a, self.n_bins.data_ptr() + 4 * a, p["n"], self.init_states.data_ptr() + ns * a,
                ctypes.addressof(p["plan"]), (p["ws"].data_ptr() + 255) // 256 * 256, p["ws_bytes"],
                self.out_off.data_ptr() + 8 * a, self.out_len.data_ptr() + 4 * a, self.status.data_ptr() + 4 * a,
                (self.final_states.data_ptr() + ns * a) if self.final_states is not None else None,
                rows_hint, self._part_counts.data_ptr() + 8 * i)
        return arr

    def settle(self):
        """After the caller has synchronised the stream of encode() / encode_chunked(): what the device reported about the run that
        was sized by a guess.  Returns {"rows": context rows the batch needs, "hint": what the run was sized by, "redone": ...}.
        The one-lane-per-slice path is exact whatever the guess; the chunked path is run again (asking the device) if the batch needed
        more rows than guessed, and its second pass is run if slices were left for it -- then the stream is synchronised again."""
        import torch
        if self._counts is None or self._hinted_path is None:
            return {"rows": 0, "hint": 0, "redone": False}
        if self._hinted_path == "parts":                     # what every part reported
            used, redone = self.rows_hint, False
            rows = max(int(self._part_counts[2 * i]) for i in range(self.n_parts))
            if used and rows > used:                         # a part needed more rows than guessed: all of them once more, asking
                self.status.copy_(self._status_before)
                self.rows_hint = 0
                self.encode_chunked()
                torch.cuda.synchronize(self.n_bins.device)
                rows, redone = max(int(self._part_counts[2 * i]) for i in range(self.n_parts)), True
            recs, rec_off = self._slice_major()
            for i, p in enumerate(self._parts):
                if int(self._part_counts[2 * i + 1]):        # slices the part left for a second pass
                    q = self._part_args[i]
                    _check(lib().avr_cabac_encode_chunked_second_pass_device(
                        self.device_index, _stream_ptr(torch), recs.data_ptr(), q.rec_off, q.n_bins, q.n_slices, q.init_states,
                        self.n_states, q.plan, q.workspace, q.workspace_bytes, self.out.data_ptr(), q.out_off, q.out_len, q.status,
                        q.final_states))
                    redone = True
            if redone:
                torch.cuda.synchronize(self.n_bins.device)
            if rows:
                self.rows_hint = min(self.n_states, rows + 8)
            return {"rows": rows, "hint": used, "redone": redone, "parts": self.n_parts}
        rows, left, used = int(self._counts[0]), int(self._counts[1]), self.rows_hint
        redone = False
        if self._hinted_path == "chunked" and used and rows > used:
            self.status.copy_(self._status_before)
            self.rows_hint = 0
            self.encode_chunked()
            torch.cuda.synchronize(self.n_bins.device)
            rows, left, redone = int(self._counts[0]), int(self._counts[1]), True
        if self._hinted_path == "chunked" and left:
            recs, rec_off = self._slice_major()
            p = self._chunk_plan()
            ws_ptr = (p["ws"].data_ptr() + 255) // 256 * 256
            _check(lib().avr_cabac_encode_chunked_second_pass_device(
                self.device_index, _stream_ptr(torch), recs.data_ptr(), rec_off.data_ptr(), self.n_bins.data_ptr(),
                self.n_slices, self.init_states.data_ptr(), self.n_states, ctypes.byref(p["plan"]), ws_ptr, p["ws_bytes"],
                self.out.data_ptr(), self.out_off.data_ptr(), self.out_len.data_ptr(), self.status.data_ptr(),
                self.final_states.data_ptr() if self.final_states is not None else None))
            torch.cuda.synchronize(self.n_bins.device)
            redone = True
        if rows:
            self.rows_hint = min(self.n_states, rows + 8)        # a little room, as avr_batch leaves (csrc/avr_api.cpp)
        return {"rows": rows, "hint": used, "redone": redone}

    def densify(self):
        """Renumber the batch onto the contexts it uses (fewer state bytes in LDS -> more waves per CU).

        Records (tiles and, if present, the slice-major copy) are rewritten in place, init_states
        is gathered to the dense numbering; final states are scattered back by final_states_full().
        Coded bytes do not change: they depend on (bin, state) sequences only."""
        import numpy as np
        import torch
        if self.kind != KIND_CABAC or getattr(self, "dense_index", None) is not None:
            return self
        L = lib()
        dev = self.n_bins.device
        sp = _stream_ptr(torch)
        bitmap = torch.zeros(32, dtype=torch.int32, device=dev)
        n_tile_recs = self.tiles.numel() // 2
        _check(L.avr_context_census_device(self.device_index, sp, self.tiles.data_ptr(), n_tile_recs, bitmap.data_ptr()))
        bits = np.unpackbits(bitmap.cpu().numpy().view(np.uint8), bitorder="little")[:1024]
        used = np.nonzero(bits)[0].astype(np.uint16)
        used = used[used < self.n_states] if self.n_states else used
        table = np.zeros(1024, dtype=np.uint16)
        table[used] = np.arange(used.size, dtype=np.uint16)
        d_table = torch.from_numpy(table.view(np.int16)).to(dev)
        d_index = torch.from_numpy(used.view(np.int16)).to(dev)
        _check(L.avr_context_remap_device(self.device_index, sp, self.tiles.data_ptr(), n_tile_recs, d_table.data_ptr()))
        if getattr(self, "rec_flat", None) is not None:
            _check(L.avr_context_remap_device(self.device_index, sp, self.rec_flat.data_ptr(),
                                              int(self.rec_off[-1]), d_table.data_ptr()))
        n_dense = max(int(used.size), 1)
        dense_init = torch.zeros(self.n_slices * n_dense, dtype=torch.uint8, device=dev)
        _check(L.avr_states_permute_device(self.device_index, sp, self.init_states.data_ptr(), self.n_states,
                                           dense_init.data_ptr(), n_dense, d_index.data_ptr(), int(used.size), self.n_slices, 0))
        torch.cuda.synchronize(dev)
        self.full_n_states, self.full_init_states = self.n_states, self.init_states
        self.dense_index, self.init_states, self.n_states = d_index, dense_init, n_dense
        self.final_states = torch.empty(self.n_slices * n_dense, dtype=torch.uint8, device=dev)
        self._plan = None
        return self

    def final_states_full(self):
        """final states in the original context numbering (contexts never touched keep their initial state)."""
        import torch
        if getattr(self, "dense_index", None) is None:
            return self.final_states
        full = self.full_init_states.clone()
        _check(lib().avr_states_permute_device(self.device_index, _stream_ptr(torch), self.final_states.data_ptr(), self.n_states,
                                               full.data_ptr(), self.full_n_states, self.dense_index.data_ptr(),
                                               self.dense_index.numel(), self.n_slices, 1))
        return full

    def _slice_major(self):
        """Slice-major records on the device (what the intra-slice parallel path reads)."""
        import torch
        if getattr(self, "rec_flat", None) is None:
            dev = self.n_bins.device
            nb = self.n_bins.to(torch.int64)
            self.rec_off = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum((nb + 7) // 8 * 8, 0)])
            self.rec_flat = torch.empty(int(self.rec_off[-1]) + 8, dtype=torch.int16, device=dev)
            _check(lib().avr_synth_generate_slices_device(
                self.device_index, _stream_ptr(torch), ctypes.byref(self.cfg), self.kind, self.n_slices,
                self.rec_off.data_ptr(), self.rec_flat.data_ptr(), None))
            if getattr(self, "dense_index", None) is not None:      # keep the copy in the batch's numbering
                import numpy as np
                table = np.zeros(1024, dtype=np.uint16)
                idx = self.dense_index.cpu().numpy().view(np.uint16)
                table[idx] = np.arange(idx.size, dtype=np.uint16)
                d_table = torch.from_numpy(table.view(np.int16)).to(dev)
                _check(lib().avr_context_remap_device(self.device_index, _stream_ptr(torch), self.rec_flat.data_ptr(),
                                                      int(self.rec_off[-1]), d_table.data_ptr()))
                torch.cuda.synchronize(dev)
        return self.rec_flat, self.rec_off

    def _plan_of(self, a, b):
        """Plan arrays and workspace of the intra-slice parallel path for slices [a, b), numbered from 0."""
        import torch
        from . import CHUNK_BINS, SORT_BLOCK_BINS, ChunkPlan
        dev = self.n_bins.device
        nb = self.n_bins[a:b].to(torch.int64)
        zero = torch.zeros(1, dtype=torch.int64, device=dev)
        ar = torch.arange(b - a, device=dev)
        res_off = torch.cat([zero, torch.cumsum((nb + 15) // 16 * 16 + 16, 0)])
        n_chunks = torch.clamp((nb + CHUNK_BINS - 1) // CHUNK_BINS, min=1)
        n_blocks = torch.clamp((nb + SORT_BLOCK_BINS - 1) // SORT_BLOCK_BINS, min=1)
        chunk_base = torch.cat([zero, torch.cumsum(n_chunks, 0)]).to(torch.int32)
        blk_base = torch.cat([zero, torch.cumsum(n_blocks, 0)]).to(torch.int32)
        dig_off = torch.cat([zero, torch.cumsum(nb // 2 + 8, 0)])
        t = dict(res_off=res_off, chunk_base=chunk_base, blk_base=blk_base, dig_off=dig_off,
                 chunk_slice=torch.repeat_interleave(ar, n_chunks).to(torch.int32),
                 blk_slice=torch.repeat_interleave(ar, n_blocks).to(torch.int32))
        plan = ChunkPlan(t["res_off"].data_ptr(), t["chunk_base"].data_ptr(), t["chunk_slice"].data_ptr(),
                         t["blk_base"].data_ptr(), t["blk_slice"].data_ptr(), t["dig_off"].data_ptr(),
                         int(res_off[-1]), int(dig_off[-1]), int(chunk_base[-1]), int(blk_base[-1]))
        p = dict(tensors=t, plan=plan, first=a, n=b - a)
        if self.kind == KIND_CABAC:
            ws_bytes = lib().avr_cabac_chunked_workspace_bytes(b - a, self.n_states, ctypes.byref(plan))
            p.update(ws_bytes=ws_bytes, ws=torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev))
        return p

    def set_parts(self, n_parts, weights=None):
        """The batch through the intra-slice parallel path as n_parts parts of consecutive slices at once, each on a stream of its
        own (avr_cabac_encode_chunked_device_parts): parts of near-equal chunk counts.  0: by the batch's size -- two parts once the
        batch is more than one round of workgroups of the path's longest kernel (what the parts fill are the part-empty last rounds:
        config 2 1.43 -> 1.37 ms in two parts, 1.49 in three; config 4 - 2 %), one part below that (128 slices of config 2: 0.67 ms in
        one, 0.70 in two)."""
        import torch
        from . import CHUNK_BINS, MAX_PARTS
        nb = self.n_bins.to(torch.int64)
        chunks = torch.clamp((nb + CHUNK_BINS - 1) // CHUNK_BINS, min=1)
        total = int(chunks.sum())
        if n_parts == 0:
            waves = (total + 63) // 64
            n_parts = 2 if (waves >= 2048 and self.n_slices >= 64) else 1
        n_parts = max(1, min(int(n_parts), MAX_PARTS, self.n_slices))
        self._parts = None
        self.n_parts = n_parts
        if n_parts > 1 and self.kind == KIND_CABAC and self._counts is not None:
            csum = torch.cumsum(chunks, 0).cpu().numpy()
            import numpy as np
            share = np.cumsum(np.asarray(weights if weights else [1.0] * n_parts, dtype=np.float64))
            share = share / share[-1]                            # (weights: the parts' shares of the chunks, for experiments; default equal)
            cuts = [0] + [int(np.searchsorted(csum, total * share[i])) + 1 for i in range(n_parts - 1)] + [self.n_slices]
            cuts = sorted(set(min(max(c, 0), self.n_slices) for c in cuts))
            self._parts = [self._plan_of(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
            self.n_parts = len(self._parts)
            self._part_counts = torch.zeros(2 * self.n_parts, dtype=torch.int32).pin_memory()
        return self.n_parts

    def _chunk_plan(self):
        """Plan arrays and workspace of the intra-slice parallel path (built once, reused)."""
        import torch
        from . import CHUNK_BINS, SORT_BLOCK_BINS, ChunkPlan
        if getattr(self, "_plan", None) is None:
            dev = self.n_bins.device
            nb = self.n_bins.to(torch.int64)
            zero = torch.zeros(1, dtype=torch.int64, device=dev)
            ar = torch.arange(self.n_slices, device=dev)
            res_off = torch.cat([zero, torch.cumsum((nb + 15) // 16 * 16 + 16, 0)])
            n_chunks = torch.clamp((nb + CHUNK_BINS - 1) // CHUNK_BINS, min=1)
            n_blocks = torch.clamp((nb + SORT_BLOCK_BINS - 1) // SORT_BLOCK_BINS, min=1)
            chunk_base = torch.cat([zero, torch.cumsum(n_chunks, 0)]).to(torch.int32)
            blk_base = torch.cat([zero, torch.cumsum(n_blocks, 0)]).to(torch.int32)
            dig_off = torch.cat([zero, torch.cumsum(nb // 2 + 8, 0)])
            t = dict(res_off=res_off, chunk_base=chunk_base, blk_base=blk_base, dig_off=dig_off,
                     chunk_slice=torch.repeat_interleave(ar, n_chunks).to(torch.int32),
                     blk_slice=torch.repeat_interleave(ar, n_blocks).to(torch.int32))
            plan = ChunkPlan(t["res_off"].data_ptr(), t["chunk_base"].data_ptr(), t["chunk_slice"].data_ptr(),
                             t["blk_base"].data_ptr(), t["blk_slice"].data_ptr(), t["dig_off"].data_ptr(),
                             int(res_off[-1]), int(dig_off[-1]), int(chunk_base[-1]), int(blk_base[-1]))
            self._plan = dict(tensors=t, plan=plan)
            if self.kind == KIND_CABAC:                     # (K2 sizes its own workspace, see encode_chunked)
                ws_bytes = lib().avr_cabac_chunked_workspace_bytes(self.n_slices, self.n_states, ctypes.byref(plan))
                self._plan.update(ws_bytes=ws_bytes, ws=torch.empty(ws_bytes + 256, dtype=torch.uint8, device=dev))
        return self._plan

    def encode_chunked(self):
        """K1 / K2 through the intra-slice parallel kernels (same bytes as encode())."""
        import torch
        recs, rec_off = self._slice_major()
        p = self._chunk_plan()
        if self.kind != KIND_CABAC:
            L = lib()
            out_total = int(self.out_off[-1].item()) if "out_total" not in p else p["out_total"]
            p["out_total"] = out_total
            if "ws_k2" not in p:
                n = L.avr_range_chunked_workspace_bytes(self.n_slices, ctypes.byref(p["plan"]), out_total)
                p["ws_k2"] = torch.empty(n + 256, dtype=torch.uint8, device=self.n_bins.device)
                p["ws_k2_bytes"] = n
            ws_ptr = (p["ws_k2"].data_ptr() + 255) // 256 * 256
            _check(L.avr_range_encode_chunked_device(
                self.device_index, _stream_ptr(torch), recs.data_ptr(), rec_off.data_ptr(), self.n_bins.data_ptr(), self.n_slices,
                ctypes.byref(p["plan"]), ws_ptr, p["ws_k2_bytes"], self.out.data_ptr(), self.out_off.data_ptr(), out_total,
                self.out_len.data_ptr(), self.status.data_ptr()))
            return
        ws_ptr = (p["ws"].data_ptr() + 255) // 256 * 256
        if self._counts is not None and getattr(self, "_parts", None):
            self._hinted_path = "parts"
            if getattr(self, "_status_before", None) is None:
                self._status_before = self.status.clone()
            self._part_args = self._part_array(self.rows_hint)
            _check(lib().avr_cabac_encode_chunked_device_parts(
                self.device_index, _stream_ptr(torch), recs.data_ptr(), self.n_states, self.out.data_ptr(),
                ctypes.cast(self._part_args, ctypes.c_void_p), self.n_parts))
            return
        if self._counts is not None:                         # sized by the previous run's context count: see encode(), settle()
            self._hinted_path = "chunked"
            if getattr(self, "_status_before", None) is None:
                self._status_before = self.status.clone()    # (a run whose guess was too small is repeated from these)
            _check(lib().avr_cabac_encode_chunked_device_hinted(
                self.device_index, _stream_ptr(torch), recs.data_ptr(), rec_off.data_ptr(), self.n_bins.data_ptr(),
                self.n_slices, self.init_states.data_ptr(), self.n_states, ctypes.byref(p["plan"]), ws_ptr, p["ws_bytes"],
                self.out.data_ptr(), self.out_off.data_ptr(), self.out_len.data_ptr(), self.status.data_ptr(),
                self.final_states.data_ptr() if self.final_states is not None else None, self.rows_hint, self._counts.data_ptr()))
            return
        _check(lib().avr_cabac_encode_chunked_device(
            self.device_index, _stream_ptr(torch), recs.data_ptr(), rec_off.data_ptr(), self.n_bins.data_ptr(),
            self.n_slices, self.init_states.data_ptr(), self.n_states, ctypes.byref(p["plan"]), ws_ptr, p["ws_bytes"],
            self.out.data_ptr(), self.out_off.data_ptr(), self.out_len.data_ptr(), self.status.data_ptr(),
            self.final_states.data_ptr() if self.final_states is not None else None))

    def resolve(self):
        """Stage 1 of K1p alone: resolved codes (uint8 tensor, slice i at res_off[i]) on the device."""
        import torch
        recs, rec_off = self._slice_major()
        p = self._chunk_plan()
        L = lib()
        if "codes" not in p:
            n = L.avr_cabac_resolve_workspace_bytes(self.n_slices, self.n_states, ctypes.byref(p["plan"]))
            p["ws1"] = torch.empty(n + 256, dtype=torch.uint8, device=self.n_bins.device)
            p["ws1_bytes"] = n
            p["codes"] = torch.empty(p["plan"].res_total + 32 + 256, dtype=torch.uint8, device=self.n_bins.device)
        al = lambda t: (t.data_ptr() + 255) // 256 * 256
        _check(L.avr_cabac_resolve_device(
            self.device_index, _stream_ptr(torch), recs.data_ptr(), rec_off.data_ptr(), self.n_bins.data_ptr(), self.n_slices,
            self.init_states.data_ptr(), self.n_states, ctypes.byref(p["plan"]), al(p["ws1"]), p["ws1_bytes"], al(p["codes"]),
            self.status.data_ptr(), self.final_states.data_ptr() if self.final_states is not None else None))
        off = al(p["codes"]) - p["codes"].data_ptr()
        return p["codes"][off:off + p["plan"].res_total + 32]

    def encode_resolved(self, codes):
        """Stage 2 of K1p alone: arithmetic coding from resolved codes (same bytes as encode())."""
        import torch
        p = self._chunk_plan()
        L = lib()
        if "ws2" not in p:
            n = L.avr_cabac_resolved_workspace_bytes(self.n_slices, ctypes.byref(p["plan"]))
            p["ws2"] = torch.empty(n + 256, dtype=torch.uint8, device=self.n_bins.device)
            p["ws2_bytes"] = n
        _check(L.avr_cabac_encode_resolved_device(
            self.device_index, _stream_ptr(torch), codes.data_ptr(), self.n_bins.data_ptr(), self.n_slices,
            ctypes.byref(p["plan"]), (p["ws2"].data_ptr() + 255) // 256 * 256, p["ws2_bytes"], self.out.data_ptr(),
            self.out_off.data_ptr(), self.out_len.data_ptr(), self.status.data_ptr()))

    def encode_codes_serial(self, codes):
        """K1 from resolved codes with one lane per slice (k_cabac_encode_codes): same bytes as encode()."""
        import torch
        p = self._chunk_plan()
        _check(lib().avr_cabac_encode_codes_device(
            self.device_index, _stream_ptr(torch), codes.data_ptr(), p["tensors"]["res_off"].data_ptr(), self.n_bins.data_ptr(),
            self.order.data_ptr(), self.n_slices, self.out.data_ptr(), self.out_off.data_ptr(), self.out_len.data_ptr(),
            self.status.data_ptr()))

    def encode_slice_major(self):
        """Same result from the slice-major layout (only for workloads built with from_host)."""
        import torch
        L = lib()
        sp = _stream_ptr(torch)
        if self.kind == KIND_CABAC:
            _check(L.avr_cabac_encode_slices_device(
                self.device_index, sp, self.rec_flat.data_ptr(), self.rec_off.data_ptr(), self.n_bins.data_ptr(),
                self.order.data_ptr(), self.n_slices, self.init_states.data_ptr(), self.n_states, self.out.data_ptr(),
                self.out_off.data_ptr(), self.out_len.data_ptr(), self.status.data_ptr(), self.final_states.data_ptr()))
        else:
            _check(L.avr_range_encode_slices_device(
                self.device_index, sp, self.rec_flat.data_ptr(), self.rec_off.data_ptr(), self.n_bins.data_ptr(),
                self.order.data_ptr(), self.n_slices, self.out.data_ptr(), self.out_off.data_ptr(),
                self.out_len.data_ptr(), self.status.data_ptr()))

    # ---- accounting (DESIGN.md "algorithmic bytes")
    def output_bytes(self) -> int:
        import torch
        return int(self.out_len.to(torch.int64).sum().item())

    def algorithmic_bytes(self) -> int:
        """2*n_bins + n_states*n_slices (K1) + out_bytes + 16*n_slices."""
        states = self.n_states * self.n_slices if self.kind == KIND_CABAC else 0
        return 2 * self.total_bins + states + self.output_bytes() + 16 * self.n_slices

    def results(self):
        """(list of bytes per slice, status list) copied to the host."""
        import torch
        torch.cuda.synchronize(self.out.device)
        out, off = self.out.cpu().numpy(), self.out_off.cpu().numpy()
        lens, st = self.out_len.cpu().numpy(), self.status.cpu().numpy()
        return [out[off[i]:off[i] + min(int(lens[i]), int(off[i + 1] - off[i]))].tobytes()
                for i in range(self.n_slices)], st.tolist()
