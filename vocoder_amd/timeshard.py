"""Time-sharded round trip: every rank owns a run of consecutive frames of every channel.

The reference processes a signal strictly in order; what it carries from one stretch of a signal
to the next is the triple of vocoder-audio/src/Vocoder/Audio.hs:31-33 — samples not yet framed,
the analysis phase and the synthesis phase.  Each of the three has a counterpart here:

* samples: rank g reads its frames' samples plus, for g > 0, the H samples before them, i.e. the
  frame that precedes its first one.  analysisStep keeps only the raw phase of the previous frame
  (Vocoder.hs:158-163), so analysing that one extra frame recovers the carried phase locally;
* synthesis phase: the chain ph <- ph + H * ph_inc (Vocoder.hs:190) is a running sum.  Each rank
  folds its own frames from zero, the per-rank totals (C*K doubles) are all-gathered, and every
  rank adds the totals of the ranks below it to the initial phase, in rank order, to obtain the
  value its chain starts from.  This is the one step that re-associates the reference's
  additions: accumulated phases agree with the sequential fold to a few ulp of their magnitude,
  not bit for bit;
* overlap-add: the last W-H output samples of rank g receive contributions from rank g+1's first
  frames.  Rank g sends that tail to rank g+1, which adds its own partial sums to it (older frames
  first, as sumFramesE folds them, Frames.hs:41-58).

The per-rank work runs through the C ABI (voc_analysis_signal_dev, voc_phase_advance_dev,
voc_synthesis_signal_dev); torch is used for the buffers and for the two exchanges.  Tensors may
live on a GPU (RCCL) or, with the host build of the kernels used by the CPU tests, in host memory
(gloo).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

from .distributed import shard_range

__all__ = ["SegmentPlan", "plan", "pieces_of", "analyse", "local_advance", "start_phase", "synthesise",
           "add_incoming_tail", "process", "process_pieces"]


@dataclass(frozen=True)
class SegmentPlan:
    rank: int
    world: int
    frames_total: int
    f_lo: int          # first frame owned
    f_hi: int          # one past the last frame owned
    has_prev: bool     # the input slice starts one hop early (frame f_lo - 1 is analysed for its phase)
    in_lo: int         # input samples needed: [in_lo, in_hi)
    in_hi: int
    out_lo: int        # output samples owned: [out_lo, out_hi)
    out_hi: int
    local_len: int     # samples the local overlap-add produces, starting at out_lo
    tail: int          # samples handed to the next rank (0 on the last rank)

    @property
    def frames(self) -> int:
        return self.f_hi - self.f_lo


def plan(par, length: int, world: int, rank: int, chunk: int | None = None) -> SegmentPlan:
    """Frame and sample ranges of `rank` for a signal of `length` samples per channel."""
    W, H = par.W, par.H
    chunk = H if chunk is None else chunk
    F = par.lib.dll.voc_num_frames(par.handle, int(length))
    f_lo, f_hi = shard_range(F, world, rank)
    tail = max(W - H, 0)
    if world > 1:
        smallest = F // world
        if smallest < 1 or smallest * H < tail:
            raise ValueError(f"{F} frames over {world} ranks: segments shorter than the overlap of {tail} samples")
    return _range_plan(par, F, f_lo, f_hi, rank > 0, rank == world - 1, chunk, rank, world)


def _range_plan(par, F: int, f_lo: int, f_hi: int, has_prev: bool, last: bool, chunk: int, rank: int = 0,
                world: int = 1) -> SegmentPlan:
    """Ranges of the run of frames [f_lo, f_hi) of a signal of F frames.  has_prev: the input
    starts one hop early; last: the run ends the signal (its output is padded to `chunk`)."""
    W, H = par.W, par.H
    in_lo = f_lo * H - (H if has_prev else 0)
    in_hi = (f_hi - 1) * H + W
    out_lo = f_lo * H
    if last:
        out_hi = par.lib.dll.voc_ola_length(par.handle, F, int(chunk))
        local_len = out_hi - out_lo
        tail_out = 0
    else:
        out_hi = f_hi * H
        local_len = max((f_hi - f_lo) * H, (f_hi - f_lo - 1) * H + W)
        tail_out = local_len - (out_hi - out_lo)
    return SegmentPlan(rank, world, F, f_lo, f_hi, has_prev, in_lo, in_hi, out_lo, out_hi, local_len, tail_out)


def pieces_of(par, sp: SegmentPlan, piece_frames: int):
    """Cuts the frames of a segment into consecutive runs of about piece_frames frames (equal to
    within one frame), none shorter than the W - H samples of overlap its successor adds onto."""
    n = max(1, -(-sp.frames // max(int(piece_frames), 1)))
    tail = max(par.W - par.H, 0)
    while n > 1 and (sp.frames // n) * par.H < tail:
        n -= 1
    out = []
    for i in range(n):
        lo, hi = shard_range(sp.frames, n, i)
        out.append((sp.f_lo + lo, sp.f_lo + hi))
    return out


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream(t):
    import torch
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream) if t.is_cuda else None


def analyse(par, sp: SegmentPlan, x_seg, ana_phase=None, volume_fix: bool = True):
    """Analysis of the rank's frames.  x_seg (C, >= in_hi - in_lo) holds samples in_lo.. of every
    channel.  ana_phase (C, K) is the phase before frame 0 and is used by rank 0 only.
    Returns (mag, ph_inc) of shape (C, frames, ldk) and the raw phase of the rank's last frame."""
    import torch

    assert x_seg.dtype == torch.float64 and x_seg.stride(-1) == 1
    c = x_seg.shape[0]
    ldk = (par.K + 1) & ~1
    mag = torch.empty((c, sp.frames, ldk), dtype=torch.float64, device=x_seg.device)
    inc = torch.empty_like(mag)
    ph = torch.zeros((c, par.K), dtype=torch.float64, device=x_seg.device)
    st = _stream(x_seg)
    if sp.has_prev:
        # one frame ahead of the segment, for its raw phase only (its magnitudes are dropped)
        scratch = torch.empty((2, c, 1, ldk), dtype=torch.float64, device=x_seg.device)
        par._call("voc_analysis_signal_dev", _ptr(x_seg), x_seg.stride(0), 1, c, _ptr(ph), _ptr(scratch[0]),
                  _ptr(scratch[1]), ldk, 0, st)
        first = x_seg[:, par.H:]
    else:
        if ana_phase is not None:
            ph.copy_(ana_phase)
        first = x_seg
    par._call("voc_analysis_signal_dev", _ptr(first), x_seg.stride(0), sp.frames, c, _ptr(ph), _ptr(mag), _ptr(inc),
              ldk, 1 if volume_fix else 0, st)
    return mag, inc, ph


def local_advance(par, inc):
    """Total phase advance of the rank's frames: the synthesis chain folded from zero."""
    import torch

    c, f, ldk = inc.shape
    total = torch.zeros((c, par.K), dtype=torch.float64, device=inc.device)
    par._call("voc_phase_advance_dev", _ptr(inc), ldk, f, c, _ptr(total), _stream(inc))
    return total


def start_phase(totals, rank: int, syn_phase=None):
    """Phase the chain of `rank` starts from: the initial phase plus the totals of ranks 0..rank-1,
    added in rank order (the same sequence on every rank)."""
    import torch

    start = torch.zeros_like(totals[0]) if syn_phase is None else syn_phase.clone()
    for r in range(rank):
        start = start + totals[r]
    return start


def synthesise(par, sp: SegmentPlan, mag, inc, start):
    """Synthesis and local overlap-add of the rank's frames.  Returns (y_local (C, local_len),
    phase after the rank's last frame)."""
    import torch

    c, f, ldk = mag.shape
    y = torch.empty((c, sp.local_len), dtype=torch.float64, device=mag.device)
    ph = start.clone()
    par._call("voc_synthesis_signal_dev", _ptr(mag), _ptr(inc), ldk, f, c, _ptr(ph), _ptr(y), y.stride(0),
              sp.local_len, _stream(mag))
    return y, ph


def add_incoming_tail(y_local, tail):
    """Adds the previous rank's tail under the first samples of this rank's output: the older
    frames' partial sums come first, then this rank's (Frames.hs:52-58 folds oldest first)."""
    n = tail.shape[1]
    if n:
        y_local[:, :n] = tail + y_local[:, :n]
    return y_local


def _through_host(t, group) -> bool:
    """gloo carries all_gather and send / recv for host tensors only: device tensors go through host
    copies under it (a rehearsal configuration; RCCL takes the device tensors as they are)."""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather(total, world, group):
    import torch
    import torch.distributed as dist
    if _through_host(total, group):
        t = total.cpu()
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t, group=group)
        return [o.to(total.device) for o in out]
    out = [torch.empty_like(total) for _ in range(world)]
    dist.all_gather(out, total, group=group)
    return out


def _neighbour_exchange(send, recv, rank, group):
    """Sends `send` (or nothing, None) to the next rank and fills `recv` (or nothing) from the previous one,
    as one batch (under RCCL the two are grouped, so neither waits for the other to be matched first)."""
    import torch
    import torch.distributed as dist

    def peer(r):
        return dist.get_global_rank(group, r) if group else r
    staged = (send is not None and _through_host(send, group)) or (recv is not None and _through_host(recv, group))
    s = send if send is None or not staged else send.cpu()
    r = recv if recv is None or not staged else torch.empty(recv.shape, dtype=recv.dtype)
    ops = []
    if s is not None:
        ops.append(dist.P2POp(dist.isend, s.contiguous(), peer(rank + 1), group))
    if r is not None:
        ops.append(dist.P2POp(dist.irecv, r, peer(rank - 1), group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if staged and recv is not None:
        recv.copy_(r)


def process(par, x_seg, length: int, chunk: int | None = None, ana_phase=None, syn_phase=None, group=None):
    """Round trip of one time segment under torch.distributed (identity filter).

    Every rank passes its slice x_seg (C, in_hi - in_lo) of the signal of `length` samples per
    channel (`plan` gives the ranges) and receives the output samples [out_lo, out_hi) it owns.
    Returns (y_owned, (ana_phase', syn_phase')) — the phases after the last frame of the whole
    signal, identical on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    sp = plan(par, length, world, rank, chunk)
    mag, inc, aph = analyse(par, sp, x_seg, ana_phase)
    total = local_advance(par, inc)
    totals = _all_gather(total, world, group) if world > 1 else [total]
    y, sph = synthesise(par, sp, mag, inc, start_phase(totals, rank, syn_phase))
    if world > 1:
        tail_in = None
        send = y[:, sp.out_hi - sp.out_lo:].contiguous() if sp.tail else None
        if rank > 0:
            prev = plan(par, length, world, rank - 1, chunk)
            if prev.tail:
                tail_in = torch.empty((y.shape[0], prev.tail), dtype=torch.float64, device=y.device)
        _neighbour_exchange(send, tail_in, rank, group)
        if tail_in is not None:
            add_incoming_tail(y, tail_in)
        last = dist.get_global_rank(group, world - 1) if group else world - 1
        dist.broadcast(aph, last, group=group)
        dist.broadcast(sph, last, group=group)
    return y[:, :sp.out_hi - sp.out_lo], (aph, sph)


def _zero_increments(filt) -> bool:
    return bool(filt is not None and getattr(filt, "zero_inc", False))


def _filtered(filt, mag, inc, first_frame: int = 0):
    """The filter chain on a run of frames that starts at frame `first_frame` of the signal."""
    if filt is None:
        return mag, inc
    if filt.changes_frame_count:
        raise ValueError("play_speed changes the number of frames: it cannot run on a time-sharded signal")
    return filt.apply_dev(mag, inc, first_frame=first_frame)


def _timed(stats, fn, device):
    """Runs fn(); with a stats dict, brackets it with device synchronisation and adds the elapsed
    seconds to stats["exchange_s"]."""
    if stats is None:
        return fn()
    import time
    import torch
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    out = fn()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    stats["exchange_s"] = stats.get("exchange_s", 0.0) + time.perf_counter() - t0
    return out


def process_pieces(par, source, length: int, piece_frames: int, sink=None, filt=None, chunk: int | None = None,
                   ana_phase=None, syn_phase=None, group=None, stats=None, comm=None):
    """Time-sharded round trip of a signal too long to hold its STFT: every rank walks its segment
    in runs of about `piece_frames` frames, so device memory holds one run's samples and STFT.

    source(lo, hi) returns samples [lo, hi) of every channel as a (C, hi - lo) float64 tensor (read
    from storage or generated in place); sink(lo, y) receives output samples [lo, lo + y.shape[1])
    of every channel — consecutive runs in order, then the rank's first W - H samples once the
    previous rank's overlap has arrived.  With sink=None the rank's output is returned as one tensor.
    filt: a `vocoder_amd.filter.BoundFilter` of per-frame stages, or None for idFilter.

    The synthesis chain of rank g starts from the phase advance of all ranks below it
    (Vocoder.hs:190 is a running sum), which is known only after those ranks have analysed and
    filtered their whole segments.  The STFT is not kept, so the segment is walked twice: once for
    the advance alone (skipped on a single rank and for chains that leave every phase increment at
    zero, Filter.hs:73), then, after the all-gather, for the output.  Within a rank the phases are
    carried from run to run exactly as the reference carries them from frame to frame; overlap-add
    across a run boundary adds the older run's partial sums first (Frames.hs:52-58), which
    re-associates that sum of at most W/H terms by one rounding.

    comm: the exchange step, for callers that do not run one process per rank under torch.distributed (ranks as
    threads of one process on one device, say): an object with `world`, `rank`, `all_gather(total) -> list of
    every rank's total in rank order`, `neighbour_exchange(send_or_None, recv_or_None)` (send goes to rank + 1,
    recv is filled from rank - 1) and `broadcast_from_last(tensors)` (in place).  None: torch.distributed.

    stats: an optional dict; the seconds spent in the two exchanges (all-gather, neighbour tails and
    the final phase broadcast, device-synchronised on both sides) are added to stats["exchange_s"].

    Returns (y or None, (ana_phase', syn_phase')) — the phases after the signal's last frame, the
    same on every rank."""
    import torch
    import torch.distributed as dist

    if comm is not None:
        world, rank = comm.world, comm.rank
    else:
        multi = dist.is_initialized() and dist.get_world_size(group) > 1
        world = dist.get_world_size(group) if multi else 1
        rank = dist.get_rank(group) if multi else 0
    chunk = par.H if chunk is None else chunk
    sp = plan(par, length, world, rank, chunk)
    F = sp.frames_total
    if F == 0:           # shorter than one frame: nothing but the (empty or padded) overlap-add of no frames
        x = source(0, 0)
        y = torch.zeros((x.shape[0], sp.out_hi), dtype=torch.float64, device=x.device)
        zero = torch.zeros((x.shape[0], par.K), dtype=torch.float64, device=x.device)
        if sink is not None and sp.out_hi:
            sink(0, y)
        return (y if sink is None else None), (zero.clone() if ana_phase is None else ana_phase.clone(),
                                               zero if syn_phase is None else syn_phase.clone())
    runs = pieces_of(par, sp, piece_frames)
    last_rank = rank == world - 1

    def run_plan(i):
        lo, hi = runs[i]
        return _range_plan(par, F, lo, hi, sp.has_prev and i == 0, last_rank and i == len(runs) - 1, chunk, rank, world)

    def analysed(i, carried):
        pp = run_plan(i)
        x = source(pp.in_lo, pp.in_hi)
        assert x.shape[1] == pp.in_hi - pp.in_lo, (x.shape, pp)
        mag, inc, ph = analyse(par, pp, x, carried)
        mag, inc = _filtered(filt, mag, inc, pp.f_lo)
        return pp, mag, inc, ph

    total = None
    if world > 1 and not _zero_increments(filt):
        carried = ana_phase
        for i in range(len(runs)):
            _, _, inc, carried = analysed(i, carried)
            if total is None:
                total = torch.zeros((inc.shape[0], par.K), dtype=torch.float64, device=inc.device)
            par._call("voc_phase_advance_dev", _ptr(inc), inc.shape[2], inc.shape[1], inc.shape[0], _ptr(total),
                      _stream(inc))
        gathered = []
        _timed(stats, lambda: gathered.extend(comm.all_gather(total) if comm is not None else _all_gather(total, world, group)),
               total.device)
        totals = gathered
    else:
        totals = None

    held = None          # the rank's first samples, which still lack the previous rank's overlap
    collected = [] if sink is None else None
    carried, sph, tail = ana_phase, None, None
    prev_tail = plan(par, length, world, rank - 1, chunk).tail if rank > 0 else 0
    for i in range(len(runs)):
        pp, mag, inc, carried = analysed(i, carried)
        if sph is None:
            if totals is not None:
                sph = start_phase(totals, rank, syn_phase)
            elif syn_phase is not None:
                sph = syn_phase.clone()
            else:
                sph = torch.zeros((mag.shape[0], par.K), dtype=torch.float64, device=mag.device)
        y, sph = synthesise(par, pp, mag, inc, sph)
        if tail is not None:
            add_incoming_tail(y, tail)
        owned = pp.out_hi - pp.out_lo
        tail = y[:, owned:].clone() if pp.tail else None
        out, lo = y[:, :owned], pp.out_lo
        if i == 0 and prev_tail:
            held, out, lo = out[:, :prev_tail].clone(), out[:, prev_tail:], lo + prev_tail
        if collected is not None:
            collected.append(out)
        elif out.shape[1]:
            sink(lo, out)
        del mag, inc, y

    if world > 1:
        tail_in = torch.empty_like(held) if prev_tail else None
        _timed(stats, lambda: (comm.neighbour_exchange(tail if sp.tail else None, tail_in) if comm is not None else
                               _neighbour_exchange(tail if sp.tail else None, tail_in, rank, group)), carried.device)
        if tail_in is not None:
            held = tail_in + held
            if collected is not None:
                collected.insert(0, held)
            else:
                sink(sp.out_lo, held)
        if comm is not None:
            _timed(stats, lambda: comm.broadcast_from_last([carried, sph]), carried.device)
        else:
            last = dist.get_global_rank(group, world - 1) if group else world - 1
            _timed(stats, lambda: (dist.broadcast(carried, last, group=group), dist.broadcast(sph, last, group=group)),
                   carried.device)
    y_all = torch.cat(collected, dim=1) if collected is not None else None
    return y_all, (carried, sph)
