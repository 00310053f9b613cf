#!/usr/bin/env python3
"""bench.py -- STFT analysis+synthesis round-trip throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the fused round trip (framing, window, R2C FFT, phase analysis, volume fix,
phase accumulation, polar synthesis, C2R IFFT, window, overlap-add) over the batch of the metric
configuration: 1024 mono streams x 60 s at 48 kHz per GPU, N = W = 4096, H = 1024, fp64, identity
filter, Hann window, synthetic signals generated in HBM before the timed region.  With N > 1 the
script runs once per GPU under torch.distributed.run -- started that way by the caller, or, when
called plainly as `python bench.py --gpus N`, by this script itself (the parent starts the ranks as
child processes before it touches a GPU and relays rank 0's line).  Streams are independent, so every
rank owns its own 1024 streams and there is no collective on the data path (weak scaling).

`--mode timeshard` times the other way of using several GPUs instead: few, long streams cut by time,
every rank owning a run of consecutive frames (64 channels x --minutes per GPU, walked in runs of
frames with the signal generated in HBM run by run), with the one exchange step of the path -- an
all-gather of the per-rank phase advance and a neighbour exchange of the overlap-add tail -- on RCCL.

`--filter` puts a frequency-domain filter between the two halves (identity: the metric, one fused call;
pitch2_lowpass1000: `--pitchShiftInterpolate 2 --lowpassBrickwall 1000`, the fourth configuration of the
baseline; envelope32: `--envelope 32`, the fifth), in both modes.

Rank 0 prints one JSON line.  `roofline` is for the slower of the two kernels of a step, its
duration measured with HIP events on the launch stream; `cpu_baseline` times the CPU oracle (a C
restatement of the reference, one core) on a bounded sample of the same workload, `cpu_baseline_config1`
on the reference's own CPU-runnable configuration in full; `parity` compares three streams of the
metric run (two of a filtered run, with the oracle's filter chain) with the oracle after the timed region.

The streams are a pure function of (seed, stream, sample index): Philox-4x32-10 keyed by the seed, the
counter holding the stream and the sample index, so a host regenerates the device's samples bit for bit
(`stream_samples_host`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_FFT, HOP, RATE, SECONDS = 4096, 1024, 48000, 60
HBM_PEAK = 8.0e12                                  # B/s, MI355X HBM3E
FP64_PEAK = 78.6e12                                # flop/s, MI355X fp64 vector
SUSTAINED_WAVE_INSTR = 4.84e11                     # v_fma_f64 wave-instructions/s the whole chip sustains (tools/ubench/fp64_sustained.hip)
FLOP_FRAME = 1.0e6                                 # ESTIMATE of the fp64 flop of one frame's round trip: 2 FFTs of 2048 complex
                                                   # points (~2.5e5) + 2049 bins x (magnitude, atan2, wrap, sin, cos) (~7e5)
K_BINS = N_FFT // 2 + 1
BYTES_ANALYSIS = 8 * HOP + 16 * K_BINS             # unique input samples in, (mag, ph_inc) out
BYTES_SYNTHESIS = 16 * K_BINS + 8 * HOP            # (mag, ph_inc) in, unique output samples out
BYTES_FRAME = BYTES_ANALYSIS + BYTES_SYNTHESIS     # 81 952 B: the algorithmic bytes of one frame


SEED = 0x5EED
PHILOX_M0, PHILOX_M1, PHILOX_W0, PHILOX_W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85


def _philox_unit(xp, c0, c1, c2, seed, u64, shr, band):
    """Philox-4x32-10 (counter (c0, c1, c2, 0), key (seed, 0)) -> 53 bits of the first two output words as a
    double in [0, 1).  One body for numpy (uint64 arrays) and torch (int64 tensors, whose wrapping products
    hold the same low 64 bits): xp supplies the array type, u64 / shr / band the three operations that differ."""
    mask = 0xFFFFFFFF
    c3 = c0 * 0
    k0, k1 = seed & mask, (seed >> 32) & mask
    for _ in range(10):
        p0, p1 = c0 * u64(PHILOX_M0), c2 * u64(PHILOX_M1)
        hi0, lo0, hi1, lo1 = band(shr(p0, 32), mask), band(p0, mask), band(shr(p1, 32), mask), band(p1, mask)
        c0, c1, c2, c3 = hi1 ^ c1 ^ u64(k0), lo1, hi0 ^ c3 ^ u64(k1), lo0
        k0, k1 = (k0 + PHILOX_W0) & mask, (k1 + PHILOX_W1) & mask
    return c0, c1


def _stream_params(s):
    """f_s = 110 * 2^((s mod 72)/12) Hz; theta_s = 2 pi * Philox(counter (0xFFFFFFFF, s, 1)): host arithmetic only."""
    s = np.asarray(s, dtype=np.uint64)
    r0, r1 = _philox_unit(np, np.full_like(s, 0xFFFFFFFF), s, np.ones_like(s), SEED, np.uint64,
                          lambda a, n: a >> np.uint64(n), lambda a, m: a & np.uint64(m))
    unit = (((r0 << np.uint64(32)) | r1) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    return 110.0 * 2.0 ** ((s % np.uint64(72)).astype(np.float64) / 12.0), unit * (2 * np.pi)


def stream_samples_host(s, lo, hi):
    """Samples [lo, hi) of stream s as the device generates them:
    x_s[n] = 0.5 sin((2 pi / 48000) f_s n + theta_s) + 1e-2 u_s[n], u_s[n] = 2 * Philox(counter (n, s, 0)) - 1."""
    f, theta = _stream_params([s])
    n = np.arange(lo, hi, dtype=np.uint64)
    r0, r1 = _philox_unit(np, n, np.full_like(n, s), np.zeros_like(n), SEED, np.uint64,
                          lambda a, k: a >> np.uint64(k), lambda a, m: a & np.uint64(m))
    u = (((r0 << np.uint64(32)) | r1) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
    # (sin is the one operation that is not the same function on host and device: the tone may differ in its
    #  last bits between the two; the noise term, which carries the generator, is identical)
    return 0.5 * np.sin((2 * np.pi / RATE) * f[0] * n.astype(np.float64) + theta[0]) + 1e-2 * (2.0 * u - 1.0)


def make_streams(c, length, dev, rank, dtype=None):
    """The metric signal, generated in HBM: streams rank*c .. rank*c + c - 1 (see stream_samples_host)."""
    s_all = np.arange(c, dtype=np.int64) + rank * c
    f_all, th_all = _stream_params(s_all)
    f_t = torch.from_numpy(f_all).to(dev)
    th_t = torch.from_numpy(th_all).to(dev)
    n = torch.arange(length, dtype=torch.int64, device=dev)
    nd = n.double()
    x = torch.empty(c, length, dtype=dtype or torch.float64, device=dev)
    step = 16
    for lo in range(0, c, step):
        hi = min(c, lo + step)
        sv = torch.from_numpy(s_all[lo:hi]).to(dev)[:, None].expand(hi - lo, length)
        r0, r1 = _philox_unit(torch, n[None, :].expand(hi - lo, length), sv, torch.zeros((), dtype=torch.int64, device=dev),
                              SEED, lambda v: v, lambda a, k: a >> k, lambda a, m: a & m)
        u = (((r0 << 21) | (r1 >> 11)) & ((1 << 53) - 1)).double() * 2.0 ** -53   # ((r0 << 32) | r1) >> 11
        x[lo:hi] = 0.5 * torch.sin((2 * np.pi / RATE) * f_t[lo:hi, None] * nd[None, :] + th_t[lo:hi, None]) + 1e-2 * (2.0 * u - 1.0)
    return x


def cpu_baseline(window, x_sample):
    """The CPU oracle (C restatement of the reference's path, -O2, no FMA contraction) on one core."""
    import oracle
    oracle.build()
    t0 = time.perf_counter()
    frames = 0
    for row in x_sample:
        frames += oracle.roundtrip(N_FFT, HOP, window, row)["nframes"]
    dt = time.perf_counter() - t0
    return frames / dt, frames, dt


def cpu_baseline_config1():
    """The reference's own CPU-runnable configuration in full (vocoder-file identity on one mono 48 kHz sine,
    60 s, N = W = 2048, hop 512, Blackman window -- the defaults of vocoder-audio/example/VocoderFile.hs:133-150
    aside): 5622 frames through the oracle on one core."""
    import oracle
    oracle.build()
    n, hop, length = 2048, 512, RATE * SECONDS
    window = oracle.make_window("blackman", n)
    x = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(length) / RATE)
    t0 = time.perf_counter()
    frames = oracle.roundtrip(n, hop, window, x)["nframes"]
    dt = time.perf_counter() - t0
    return frames / dt, frames, dt


def make_filter(name, par):
    """The filter chains of the baseline's configurations, bound to the context (None: idFilter)."""
    from vocoder_amd import filter as Fl
    if name == "identity":
        return None
    if name == "pitch2_lowpass1000":      # --pitchShiftInterpolate 2 --lowpassBrickwall 1000 (VocoderFile.hs:74-125)
        chain = Fl.pitch_shift_interpolate(2.0) >> Fl.lowpass_brickwall(1000.0)
    elif name == "envelope32":            # --envelope 32
        chain = Fl.envelope_filter(32)
    else:
        raise SystemExit(f"unknown filter {name}")
    return chain.bind(par, float(RATE) / par.N)


def parity_check(par, win, x, y, streams=3, bar=1e-9):
    """Streams 0..streams-1 of a metric step that started from zero phases against the oracle: relative L2 of the
    output samples (bar 1e-9; the single-precision variant is held to its own, looser bar against the same
    double-precision oracle), and the leftover-free length check."""
    import oracle
    oracle.build()
    worst, frames = 0.0, 0
    for s_ in range(streams):
        r = oracle.roundtrip(N_FFT, HOP, win, x[s_].double().cpu().numpy())
        got = y[s_].double().cpu().numpy()
        if got.shape != r["y"].shape:
            return {"ok": False, "error": f"stream {s_}: {got.shape} samples, oracle {r['y'].shape}"}
        worst = max(worst, float(np.linalg.norm(got - r["y"]) / np.linalg.norm(r["y"])))
        frames += int(r["nframes"])
    return {"ok": bool(worst <= bar), "rel_l2_max": worst, "bar": bar, "streams": streams, "frames": frames,
            "against": "CPU oracle (C restatement of the reference path) on the same samples, zero initial phases"}


ORACLE_CHAINS = {"pitch2_lowpass1000": [("pitch_shift", 2.0), ("lowpass_brickwall", 1000.0)], "envelope32": [("envelope", 32)]}


def parity_check_filtered(name, win, x, y, streams=2):
    """The filtered configurations against the reference pipeline assembled from oracle pieces (analysis with volumeFix,
    the filter chain of oracle/filters_oracle.c, synthesisStage, sumFramesE; Conduit.hs:46-47) on the first streams."""
    import oracle
    oracle.build()
    worst, frames = 0.0, 0
    k = N_FFT // 2 + 1
    for s_ in range(streams):
        rt = oracle.roundtrip(N_FFT, HOP, win, x[s_].double().cpu().numpy(), HOP, want_stft=True)
        mag, inc = oracle.apply_filter(float(RATE) / N_FFT, ORACLE_CHAINS[name], rt["mag"], rt["ph_inc"])
        _, fr = oracle.synthesis_stage(N_FFT, HOP, win, np.zeros(k), mag, inc)
        ref = np.zeros(oracle.ola_length(mag.shape[0], N_FFT, HOP, HOP))
        for j in range(mag.shape[0]):
            ref[j * HOP:j * HOP + N_FFT] = ref[j * HOP:j * HOP + N_FFT] + fr[j]
        got = y[s_].double().cpu().numpy()
        if got.shape != ref.shape:
            return {"ok": False, "error": f"stream {s_}: {got.shape} samples, oracle {ref.shape}"}
        worst = max(worst, float(np.linalg.norm(got - ref) / np.linalg.norm(ref)))
        frames += int(mag.shape[0])
    return {"ok": bool(worst <= 1e-9), "rel_l2_max": worst, "bar": 1e-9, "streams": streams, "frames": frames,
            "against": "CPU oracle pipeline with the same filter chain on the same samples, zero initial phases"}


def cpu_baseline_all_cores(window, x_sample, cores):
    """The same oracle with the streams dealt over `cores` threads (the C call releases the GIL):
    what a host would reach by running one single-threaded reference process per core."""
    import oracle
    from concurrent.futures import ThreadPoolExecutor
    oracle.build()

    def work(rows):
        return sum(oracle.roundtrip(N_FFT, HOP, window, row)["nframes"] for row in rows)

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        frames = sum(ex.map(work, [x_sample[i::cores] for i in range(cores)]))
    dt = time.perf_counter() - t0
    return frames / dt, frames, dt


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` called plainly: start the N ranks as child processes of a parent that
    has not touched a GPU and exit with the launcher's status.  Rank 0's JSON line is the only thing
    that reaches stdout; whatever else the ranks or the communication library print there goes to stderr."""
    import subprocess
    # --standalone: the launcher's own rendezvous picks the port and keeps it (no window between choosing a
    # free port and binding it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in child.stdout:
        is_result = line.startswith("{") and '"metric"' in line
        (sys.stdout if is_result else sys.stderr).write(line)
        (sys.stdout if is_result else sys.stderr).flush()
    raise SystemExit(child.wait())


def timeshard_mode(args, world, rank, dev, red_dev, D):
    """Few long streams cut by time over the ranks (weak scaling: every rank owns --minutes of signal)."""
    from vocoder_amd import vocoder as V, timeshard as T, window as Wn
    par = V.VocoderParams(N_FFT, HOP, Wn.hann_window(N_FFT), device=dev.index)
    filt = make_filter(args.filter, par)
    c = args.channels
    length = int(args.minutes * 60 * RATE) * world
    freqs = (110.0 * 2.0 ** ((torch.arange(c, device=dev) % 72) / 12.0)).double()[:, None]

    def generate(lo, hi):
        t = torch.arange(lo, hi, device=dev, dtype=torch.float64)[None, :]
        return 0.5 * torch.sin((2 * np.pi / RATE) * freqs * t)

    # The rank's stretch of the signal is resident in device memory before the timed region starts (the runs of
    # the warm-up walk fill the cache; the timed walks ask for the same ranges) unless it is too long to hold:
    # then every run is generated where it is used, inside the timed region, and the line says so.
    resident = c * (length // world) * 8 <= 64 << 30
    cache = {}

    def source(lo, hi):
        if not resident:
            return generate(lo, hi)
        if (lo, hi) not in cache:
            cache[(lo, hi)] = generate(lo, hi)
        return cache[(lo, hi)]

    acc = torch.zeros((), dtype=torch.float64, device=dev)
    timed = [False]

    def sink(lo, y):                                    # the output samples are in device memory when this is called
        if not timed[0]:
            acc.add_(y.square().sum())                  # (warm-up walks: a checksum, so that the output is looked at once)

    def fence():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    stats = {}
    for _ in range(max(args.warmup, 1)):
        T.process_pieces(par, source, length, args.piece, sink=sink, filt=filt)
    assert bool(torch.isfinite(acc)) and float(acc) > 0.0
    timed[0] = True
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        T.process_pieces(par, source, length, args.piece, sink=sink, filt=filt, stats=stats)
    fence()
    dt = D.max_over_ranks(time.perf_counter() - t0, red_dev)
    exch = D.max_over_ranks(stats.get("exchange_s", 0.0), red_dev)
    if rank == 0:
        frames = par.lib.dll.voc_num_frames(par.handle, length) * c
        print(json.dumps({
            "metric": "STFT frames/sec analysis+synthesis (N=4096,hop=1024,fp64), long streams cut by time",
            "value": frames * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{c} channels x {args.minutes} min @48 kHz per GPU, N=W=4096 hop=1024 fp64 "
                                   f"round-trip (Hann), filter {args.filter}, cut by time over {world} rank(s), "
                                   f"runs of {args.piece} frames, input "
                                   + ("resident in device memory" if resident else "generated run by run inside the timed region"),
                       "filter": args.filter,
                       "sharding": "by time segment: halo frame, all-gather of the phase advance, neighbour tail exchange",
                       # chains that leave every increment at zero (envelope) need no walk for the phase advance
                       "walks_per_step": 2 if (world > 1 and args.filter != "envelope32") else 1},
            "exchange_ms_per_step": exch / args.steps * 1e3,
        }), flush=True)
    D.finish()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=1024, help="streams per GPU (metric configuration: 1024)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle comparison of three streams after the timed region")
    ap.add_argument("--precision", choices=("f64", "f32"), default="f64",
                    help="f32: the single-precision variant (not the metric; reported with dtype f32)")
    ap.add_argument("--cpu-streams", type=int, default=24, help="streams of the workload the CPU oracle is timed on")
    ap.add_argument("--backend", default="nccl", help="process-group backend (gloo: rehearsal of the multi-rank path without RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a one-GPU box: every rank uses device 0")
    ap.add_argument("--mode", choices=("streams", "timeshard"), default="streams",
                    help="streams: the metric (streams sharded over GPUs); timeshard: long streams cut by time")
    ap.add_argument("--filter", choices=("identity", "pitch2_lowpass1000", "envelope32"), default="identity",
                    help="frequency-domain filter between analysis and synthesis (identity: the metric)")
    ap.add_argument("--channels", type=int, default=64, help="timeshard: channels of the long signal")
    ap.add_argument("--minutes", type=float, default=10.0, help="timeshard: minutes of signal per GPU")
    ap.add_argument("--piece", type=int, default=8192, help="timeshard: frames per run")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)          # before anything touches a GPU in this process

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    from vocoder_amd import distributed as D
    D.init(args.backend, dev)
    red_dev = dev if args.backend == "nccl" else "cpu"   # where the one reduced scalar lives
    if args.mode == "timeshard":
        return timeshard_mode(args, world, rank, dev, red_dev, D)

    from vocoder_amd import vocoder as V, conduit as Cd, window as Wn
    win = Wn.hann_window(N_FFT)
    f32 = args.precision == "f32"
    tdt = torch.float32 if f32 else torch.float64
    par = V.VocoderParams(N_FFT, HOP, win, device=local_rank, precision=args.precision)
    c, length = args.streams, RATE * SECONDS
    frames_per_stream = Cd.num_frames(par, length)
    frames = c * frames_per_stream
    x = make_streams(c, length, dev, rank, tdt)
    aph = torch.zeros(c, par.K, dtype=tdt, device=dev)
    sph = torch.zeros_like(aph)
    if f32 and args.filter != "identity":
        raise SystemExit("--filter with --precision f32 is not a bench configuration")
    filt = make_filter(args.filter, par)
    y = ws = None
    if filt is None:   # the metric: one fused call, its STFT in a workspace allocated once
        y = torch.empty(c, Cd.ola_length(par, frames_per_stream, HOP), dtype=tdt, device=dev)
        ws = torch.empty(Cd.roundtrip_workspace_bytes(par, length, c), dtype=torch.uint8, device=dev)

    def step():
        if filt is None:
            Cd.process_dev(par, x, y, aph, sph, workspace=ws)
        else:   # analysis .| volumeFix .| filter .| synthesis, the STFT in HBM between them (Conduit.hs:46-47); its
            # tensors come from torch's caching allocator: no device allocation after the warm-up steps
            return Cd.process_filter_dev(par, filt, x, ana_phase=aph, syn_phase=sph)[0]

    def fence():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        aph.zero_(); sph.zero_()
        step()
    fence()
    par.lib.check(par.lib.dll.voc_profile_enable(par.handle, 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ms = (np.zeros(4), np.zeros(4, dtype=np.int64))
    par.lib.check(par.lib.dll.voc_profile_read(par.handle, ms[0].ctypes.data, ms[1].ctypes.data))
    par.lib.check(par.lib.dll.voc_profile_enable(par.handle, 0))
    dt = D.max_over_ranks(dt, red_dev)

    if rank == 0:
        value = frames * world * args.steps / dt
        names = ("analysis_kernel", "phase_carry_kernel", "synthesis_kernel", "filter_and_other_kernels")
        avg = [ms[0][i] / ms[1][i] if ms[1][i] else 0.0 for i in range(4)]     # ms per launch
        dom = 0 if avg[0] >= avg[2] else 2
        # bytes one launch moves: the library may cut a step's frames into several launches of a kernel
        scale = 0.5 if f32 else 1.0                                            # float buffers: half the bytes
        alg = scale * (BYTES_ANALYSIS, 0, BYTES_SYNTHESIS)[dom] * frames * args.steps / max(int(ms[1][dom]), 1)
        achieved = alg / (avg[dom] * 1e-3) / 1e9 if avg[dom] else 0.0
        # a filter pass reads and writes the STFT once more: 32 K bytes per frame on top of the path's own
        bytes_frame = BYTES_FRAME + (32 * K_BINS if filt is not None else 0)
        metric = "STFT frames/sec analysis+synthesis (N=4096,hop=1024,fp64); % HBM roofline"
        if f32:
            metric = "STFT frames/sec analysis+synthesis (N=4096,hop=1024), single-precision variant; % HBM roofline"
        workload = (f"{c} mono synthetic streams x 60 s @48 kHz per GPU, N=W=4096 hop=1024 "
                    f"{'fp32' if f32 else 'fp64'} identity round-trip (Hann), configs[1]")
        if filt is not None:
            metric = f"STFT frames/sec analysis+filter+synthesis (N=4096,hop=1024,fp64), filter {args.filter}; % HBM roofline"
            workload = (f"{c} synthetic channels x 60 s @48 kHz per GPU, N=W=4096 hop=1024 fp64 round-trip (Hann) with "
                        f"filter {args.filter} between analysis and synthesis" +
                        (", configs[3]" if args.filter == "pitch2_lowpass1000" else ""))
        out = {
            "metric": metric,
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": workload,
                       "streams_per_gpu": c, "frames_per_step_per_gpu": frames, "sharding": "by stream, no collective"},
            "hbm_roofline_frac_whole_path": value / world * scale * bytes_frame / HBM_PEAK,
            "kernel_ms": {names[i]: avg[i] for i in range(4) if ms[1][i]},
            "kernel_launches_per_step": {names[i]: int(ms[1][i]) // args.steps for i in range(4) if ms[1][i]},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / (HBM_PEAK / 1e9), "traffic": None},
        }
        if filt is not None:
            out["config"]["filter"] = args.filter
            out["config"]["algorithmic_bytes_per_frame"] = bytes_frame
        # PMC-derived figures are not measured in this run: they come from the committed counter passes of
        # the same command (tools/profile_bench.sh) and say so
        tfile = os.path.join(ROOT, "profiles", "traffic.json")   # HBM bytes per launch
        if os.path.exists(tfile):
            try:
                t = json.load(open(tfile))
                if t.get("streams_per_gpu") == args.streams and not f32 and filt is None:   # measured on this workload only
                    out["roofline"]["traffic"] = t.get(names[dom])
                    out["roofline"]["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc passes over this command)"
            except Exception:
                pass
        # The kernels sit past the ridge: vector instruction issue, not HBM, is what they saturate.  Reported beside
        # the HBM figure, from the committed counter passes: vector instructions per frame, the rate they are
        # issued at in this run, and that rate against what the chip sustains on fp64 multiply-adds alone
        # (profiles/r03_fp64_sustained.txt: 4.84e11 wave-instructions/s = 62 TFLOP/s, below the 78.6 of the
        # data sheet because the clock settles near 1.9 GHz under that load).  The flop count is an estimate
        # and enters no fraction.
        if not f32 and filt is None:
            fp = {"datasheet_peak_tflops": FP64_PEAK / 1e12, "sustained_fma_wave_instr_per_s": SUSTAINED_WAVE_INSTR,
                  "flop_per_frame_estimate": FLOP_FRAME, "tflops_estimate": value / world * FLOP_FRAME / 1e12}
            vfile = os.path.join(ROOT, "profiles", "valu.json")
            if os.path.exists(vfile):
                try:
                    v = json.load(open(vfile))
                    if v.get("streams_per_gpu") == args.streams:
                        per_frame = v.get("valu_wave_instr_per_frame")
                        fp["instr_per_wave_frame"] = v.get("instr_per_wave_frame")
                        fp["valu_busy_frac"] = v.get("valu_busy_frac")
                        fp["lds_busy_frac"] = v.get("lds_busy_frac")
                        if per_frame:
                            fp["valu_wave_instr_per_frame"] = per_frame
                            fp["achieved_wave_instr_per_s"] = value / world * per_frame
                            fp["frac"] = value / world * per_frame / SUSTAINED_WAVE_INSTR
                        fp["counters_source"] = "profiles/valu.json (rocprofv3 --pmc passes over this command)"
                except Exception:
                    pass
            out["roofline"]["fp64_valu"] = fp
        if world == 1 and filt is None and not args.no_parity:
            aph.zero_(); sph.zero_()
            step()
            torch.cuda.synchronize()
            # (single precision: 60 s of the metric signal stay within 1e-3 of the double-precision oracle, the
            #  float increments' rounding random-walking in the noise part; measured 2.8e-4)
            out["parity"] = parity_check(par, win, x, y, bar=1e-3 if f32 else 1e-9)
        if world == 1 and filt is not None and not args.no_parity:
            aph.zero_(); sph.zero_()
            yf = step()
            torch.cuda.synchronize()
            out["parity"] = parity_check_filtered(args.filter, win, x, yf)
        if world == 1 and not args.no_cpu_baseline and filt is None:
            ncpu = min(args.cpu_streams, c)
            v, nfr, secs = cpu_baseline(win, x[:ncpu].double().cpu().numpy())
            out["cpu_baseline"] = {"value": v, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": f"{ncpu} of the {c} streams x 60 s ({nfr} frames, {secs:.1f} s of CPU time), "
                                             "C restatement of the reference path"}
            v1, nfr1, secs1 = cpu_baseline_config1()
            out["cpu_baseline_config1"] = {"value": v1, "unit": "frames/s", "cores": 1, "kind": "port",
                                           "sample": f"configs[0] in full: 1 mono 440 Hz sine x 60 s @48 kHz, N=W=2048 hop=512 "
                                                     f"Blackman, identity ({nfr1} frames, {secs1:.2f} s of CPU time)"}
            cores = min(16, len(os.sched_getaffinity(0)))       # a one-GPU box's share of the host
            if cores > 1:
                nall = min(4 * cores, c)
                v, nfr, secs = cpu_baseline_all_cores(win, x[:nall].double().cpu().numpy(), cores)
                out["cpu_baseline_all_cores"] = {"value": v, "unit": "frames/s", "cores": cores, "kind": "port",
                                                 "sample": f"{nall} of the {c} streams x 60 s over {cores} threads "
                                                           f"({nfr} frames, {secs:.1f} s wall)"}
        print(json.dumps(out), flush=True)
    D.finish()


if __name__ == "__main__":
    main()
