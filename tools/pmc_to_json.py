"""Turns the per-kernel counter means of tools/profile_bench.sh (gpurun_out/profile/pmc.json) into the two
small files bench.py quotes beside its own measurements:

  profiles/traffic.json  HBM bytes per launch of each kernel: 2 * FETCH_SIZE + WRITE_SIZE, both counters in
                         KiB; FETCH_SIZE doubled because gfx950 tallies a 128-byte read request as 64 bytes
  profiles/valu.json     vector instructions per wave and frame (SQ_INSTS_VALU / (waves per frame * frames per
                         launch)), per frame, and the fraction of cycles the vector ALUs are busy
                         (SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs, over GRBM_GUI_ACTIVE / 8 XCDs of the SAME pass)

usage: python tools/pmc_to_json.py gpurun_out/profile/pmc.json [--streams 1024] [--frames-per-stream 2809]
"""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMDS, XCDS, WAVES_PER_GROUP = 1024, 8, 4


def pick(means, prefix):
    hits = [v for k, v in means.items() if k.startswith(prefix)]
    if not hits:
        return None
    return max(hits, key=lambda d: d.get("SQ_INSTS_VALU", d.get("FETCH_SIZE", d.get("WRITE_SIZE", 0))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("pmc")
    ap.add_argument("--streams", type=int, default=1024)
    ap.add_argument("--frames-per-stream", type=int, default=2809)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles"))
    a = ap.parse_args()
    means = json.load(open(a.pmc))
    kernels = {"analysis_kernel": "analysis_fast", "phase_carry_kernel": "phase_carry_kernel",
               "synthesis_kernel": "synthesis_fast"}
    frames = a.streams * a.frames_per_stream
    traffic = {"_note": "HBM bytes per launch on the metric configuration: 2*FETCH_SIZE + WRITE_SIZE, counters in KiB "
                        "from separate rocprofv3 --pmc passes over bench.py; FETCH_SIZE is doubled because gfx950 "
                        "tallies 128-byte read requests at 64 bytes.",
               "streams_per_gpu": a.streams}
    valu = {"_note": "From rocprofv3 --pmc passes over bench.py. instr_per_wave_frame = SQ_INSTS_VALU / (waves per "
                     "frame * frames per launch); valu_wave_instr_per_frame = sum over the two kernels of SQ_INSTS_VALU / "
                     "frames; valu_busy_frac = (SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs), both "
                     "counters from the same pass; lds_busy_frac likewise from SQ_ACTIVE_INST_LDS and its own pass.",
            "streams_per_gpu": a.streams, "instr_per_wave_frame": {}, "valu_busy_frac": {}, "lds_busy_frac": {},
            "lds_bank_conflict_frac": {}, "kernel_names": {}, "waves_per_frame": {}, "lds_instr_per_launch": {}}
    per_frame_total = 0.0
    for short, prefix in kernels.items():
        d = pick(means, prefix)
        if d is None:
            continue
        if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
            traffic[short] = int((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024)
        if "SQ_INSTS_VALU" in d and short != "phase_carry_kernel":
            # waves per frame: the sixteen-point kernels run a frame of N = 4096 on two waves, the eight-point ones on four
            name = next(k for k, v in means.items() if v is d)
            waves = 2 if "fast16" in name else WAVES_PER_GROUP
            valu["kernel_names"][short] = name
            valu["waves_per_frame"][short] = waves
            valu["instr_per_wave_frame"][short] = round(d["SQ_INSTS_VALU"] / (waves * frames), 1)
            per_frame_total += d["SQ_INSTS_VALU"] / frames

            def busy(counter):   # every counter against the clock of the pass it was collected in
                cycles = d.get(counter + "@GRBM_GUI_ACTIVE", d["GRBM_GUI_ACTIVE"]) / XCDS
                return round(d[counter] * 4 / SIMDS / cycles, 3)
            valu["valu_busy_frac"][short] = busy("SQ_ACTIVE_INST_VALU")
            if "SQ_ACTIVE_INST_LDS" in d:
                valu["lds_busy_frac"][short] = busy("SQ_ACTIVE_INST_LDS")
            if "SQ_INSTS_LDS" in d:
                valu["lds_instr_per_launch"][short] = d["SQ_INSTS_LDS"]
            if "SQ_LDS_BANK_CONFLICT" in d and d.get("SQ_LDS_IDX_ACTIVE"):
                valu["lds_bank_conflict_frac"][short] = round(d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], 3)
    if per_frame_total:
        valu["valu_wave_instr_per_frame"] = round(per_frame_total, 1)
    json.dump(traffic, open(os.path.join(a.out, "traffic.json"), "w"), indent=1)
    json.dump(valu, open(os.path.join(a.out, "valu.json"), "w"), indent=1)
    print(json.dumps(traffic, indent=1))
    print(json.dumps(valu, indent=1))


if __name__ == "__main__":
    main()
