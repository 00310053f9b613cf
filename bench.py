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

Rank 0 prints one JSON line.  `roofline` is for the slower of the two kernels of a step, its
duration measured with HIP events on the launch stream; `cpu_baseline` times the CPU oracle (a C
restatement of the reference, one core) on a bounded sample of the same workload.
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
FLOP_FRAME = 1.0e6                                 # fp64 flop of one frame's round trip: 2 FFTs of 2048 complex points
                                                   # (~2.5e5) + 2049 bins x (magnitude, atan2, wrap, sin, cos) (~7e5)
K_BINS = N_FFT // 2 + 1
BYTES_ANALYSIS = 8 * HOP + 16 * K_BINS             # unique input samples in, (mag, ph_inc) out
BYTES_SYNTHESIS = 16 * K_BINS + 8 * HOP            # (mag, ph_inc) in, unique output samples out
BYTES_FRAME = BYTES_ANALYSIS + BYTES_SYNTHESIS     # 81 952 B: the algorithmic bytes of one frame


def make_streams(c, length, dev, rank, dtype=None):
    """x_s[n] = 0.5 sin(2 pi f_s n / 48000 + theta_s) + 1e-2 u_s[n], f_s = 110 * 2^((s mod 72)/12) Hz."""
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED + rank)
    n = torch.arange(length, dtype=torch.float64, device=dev)
    s = torch.arange(c, device=dev) + rank * c
    f = 110.0 * torch.pow(2.0, (s % 72).double() / 12.0)
    theta = torch.rand(c, generator=g, dtype=torch.float64, device=dev) * (2 * np.pi)
    x = torch.empty(c, length, dtype=dtype or torch.float64, device=dev)
    for lo in range(0, c, 32):
        hi = min(c, lo + 32)
        x[lo:hi] = 0.5 * torch.sin((2 * np.pi / RATE) * f[lo:hi, None] * n[None, :] + theta[lo:hi, None])
        x[lo:hi] += 1e-2 * (2 * torch.rand(hi - lo, length, generator=g, dtype=torch.float64, device=dev) - 1)
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
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
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
    c = args.channels
    length = int(args.minutes * 60 * RATE) * world
    freqs = (110.0 * 2.0 ** ((torch.arange(c, device=dev) % 72) / 12.0)).double()[:, None]

    def source(lo, hi):                                 # generated where it is used: the whole signal is never resident
        t = torch.arange(lo, hi, device=dev, dtype=torch.float64)[None, :]
        return 0.5 * torch.sin((2 * np.pi / RATE) * freqs * t)

    acc = torch.zeros((), dtype=torch.float64, device=dev)

    def sink(lo, y):
        acc.add_(y.square().sum())

    def fence():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    stats = {}
    for _ in range(args.warmup):
        T.process_pieces(par, source, length, args.piece, sink=sink)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        T.process_pieces(par, source, length, args.piece, sink=sink, stats=stats)
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
            "config": {"workload": f"{c} channels x {args.minutes} min @48 kHz per GPU, N=W=4096 hop=1024 fp64 identity "
                                   f"round-trip (Hann), cut by time over {world} rank(s), runs of {args.piece} frames",
                       "sharding": "by time segment: halo frame, all-gather of the phase advance, neighbour tail exchange",
                       "walks_per_step": 2 if world > 1 else 1},
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
    ap.add_argument("--precision", choices=("f64", "f32"), default="f64",
                    help="f32: the single-precision variant (not the metric; reported with dtype f32)")
    ap.add_argument("--cpu-streams", type=int, default=24, help="streams of the workload the CPU oracle is timed on")
    ap.add_argument("--backend", default="nccl", help="process-group backend (gloo: rehearsal of the multi-rank path without RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a one-GPU box: every rank uses device 0")
    ap.add_argument("--mode", choices=("streams", "timeshard"), default="streams",
                    help="streams: the metric (streams sharded over GPUs); timeshard: long streams cut by time")
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
    y = torch.empty(c, Cd.ola_length(par, frames_per_stream, HOP), dtype=tdt, device=dev)
    aph = torch.zeros(c, par.K, dtype=tdt, device=dev)
    sph = torch.zeros_like(aph)
    ws = torch.empty(Cd.roundtrip_workspace_bytes(par, length, c), dtype=torch.uint8, device=dev)

    def step():
        Cd.process_dev(par, x, y, aph, sph, workspace=ws)

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
        names = ("analysis_kernel", "phase_carry_kernel", "synthesis_kernel")
        avg = [ms[0][i] / ms[1][i] if ms[1][i] else 0.0 for i in range(3)]     # ms per launch
        dom = 0 if avg[0] >= avg[2] else 2
        # bytes one launch moves: the library may cut a step's frames into several launches of a kernel
        scale = 0.5 if f32 else 1.0                                            # float buffers: half the bytes
        alg = scale * (BYTES_ANALYSIS, 0, BYTES_SYNTHESIS)[dom] * frames * args.steps / max(int(ms[1][dom]), 1)
        achieved = alg / (avg[dom] * 1e-3) / 1e9 if avg[dom] else 0.0
        out = {
            "metric": "STFT frames/sec analysis+synthesis (N=4096,hop=1024,fp64); % HBM roofline" if not f32 else
                      "STFT frames/sec analysis+synthesis (N=4096,hop=1024), single-precision variant; % HBM roofline",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"{c} mono synthetic streams x 60 s @48 kHz per GPU, N=W=4096 hop=1024 "
                                   f"{'fp32' if f32 else 'fp64'} identity round-trip (Hann), configs[1]",
                       "streams_per_gpu": c, "frames_per_step_per_gpu": frames, "sharding": "by stream, no collective"},
            "hbm_roofline_frac_whole_path": value / world * scale * BYTES_FRAME / HBM_PEAK,
            "kernel_ms": {names[i]: avg[i] for i in range(3) if ms[1][i]},
            "kernel_launches_per_step": {names[i]: int(ms[1][i]) // args.steps for i in range(3) if ms[1][i]},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK / 1e9,
                         "unit": "GB/s", "frac": achieved / (HBM_PEAK / 1e9), "traffic": None},
        }
        # PMC-derived figures are not measured in this run: they come from the committed counter passes of
        # the same command (tools/profile_bench.sh) and say so
        tfile = os.path.join(ROOT, "profiles", "traffic.json")   # HBM bytes per launch
        if os.path.exists(tfile):
            try:
                t = json.load(open(tfile))
                if t.get("streams_per_gpu") == args.streams and not f32:   # measured on this workload only
                    out["roofline"]["traffic"] = t.get(names[dom])
                    out["roofline"]["traffic_source"] = "profiles/traffic.json (rocprofv3 --pmc passes over this command)"
            except Exception:
                pass
        # The kernels sit past the ridge: fp64 vector issue, not HBM, is what they saturate.  Reported
        # beside the HBM figure: the whole path against the fp64 vector peak (algorithmic flop count),
        # and, from the counter passes, vector instructions per wave and frame and the busy fraction.
        if not f32:
            fp = {"peak": FP64_PEAK / 1e12, "unit": "TFLOP/s", "flop_per_frame": FLOP_FRAME,
                  "achieved": value / world * FLOP_FRAME / 1e12, "frac": value / world * FLOP_FRAME / FP64_PEAK}
            vfile = os.path.join(ROOT, "profiles", "valu.json")
            if os.path.exists(vfile):
                try:
                    v = json.load(open(vfile))
                    if v.get("streams_per_gpu") == args.streams:
                        fp["instr_per_wave_frame"] = v.get("instr_per_wave_frame")
                        fp["valu_busy_frac"] = v.get("valu_busy_frac")
                        fp["counters_source"] = "profiles/valu.json (rocprofv3 --pmc passes over this command)"
                except Exception:
                    pass
            out["roofline"]["fp64_valu"] = fp
        if world == 1 and not args.no_cpu_baseline:
            ncpu = min(args.cpu_streams, c)
            v, nfr, secs = cpu_baseline(win, x[:ncpu].double().cpu().numpy())
            out["cpu_baseline"] = {"value": v, "unit": "frames/s", "cores": 1, "kind": "port",
                                   "sample": f"{ncpu} of the {c} streams x 60 s ({nfr} frames, {secs:.1f} s of CPU time), "
                                             "C restatement of the reference path"}
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
