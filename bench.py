#!/usr/bin/env python3
"""
bench.py -- headline benchmark: windows/sec end-to-end (dist -> Rips H0/H1 -> Wasserstein),
47-channel EEG, on N MI355X (BASELINE.json `metric`).

A step = ONE pass of the per-window hot path (tda_eeg_audio_amd/pipeline.py::run_step, the
batched counterpart of scripts/tda_eeg_audio_comparison.py:77-122) over one batch of synthetic
input already resident in HBM: per window corr->dist, Rips(EEG 47x47), Takens+Rips(audio),
Wasserstein H0 and H1, H1 features, and the per-recording reductions.  Workload at N=1 =
BASELINE.json configs[1]: 710 EEG windows of one band (+ the 710 matching audio windows).
N>1: every rank gets its own 710 windows (weak scaling); the only collective is one all-gather
of the per-recording result rows (RCCL over xGMI), inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (largest share of the
step), with its average launch duration measured live with events on the launch stream.
`cpu_baseline` times the CPU oracle (oracle/tda_oracle.c, kind "port", 1 core) on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ALGORITHMIC HBM bytes per unit of work of each kernel
ALG_BYTES = {
    "corr_dist": 47 * 250 * 8 + 47 * 47 * 8,          # window in, distance matrix out (no corr copy)
    "rips_eeg": 47 * 47 * 8 + 1100,                   # 18.7 KB/window from the stored f64 matrix
    "tau": 250 * 8 + 4,
    "rips_audio": 250 * 8 + 4 + 1500,                 # ~3.5 KB/window
    "wasserstein_h0": 2700 + 8,                       # <= 2.7 KB/pair
    "wasserstein_h1": 2700 + 8,
    "features_eeg": 2 * (1100 + 88) + 2 * 22 * 8,
    "features_audio": 1100 + 88,
    "reduce": 2 * 8,
}
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--windows", type=int, default=710, help="windows per GPU per step (configs[1] = 710)")
    ap.add_argument("--windows-per-recording", type=int, default=15, help="cmp:39 MAX_WINDOWS")
    ap.add_argument("--band", default="beta")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-defer", action="store_true",
                    help="launch the widening passes of the Rips stages with every step instead of verify-then-publish")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("TDA_LANES", "3")),
                    help="batches in flight (pipeline.Lanes): 1 = strictly one step after the other")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 and the all-gather goes through gloo "
                         "(exercises the N>1 code path on a one-GPU box; numbers are meaningless)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tda_eeg_audio_amd import _lib, pipeline, synth
    from tda_eeg_audio_amd import dist as tdist

    if args.share_gpu:
        os.environ["LOCAL_RANK_REAL"] = os.environ.get("LOCAL_RANK", "0")
    rank, world, local = tdist.init_from_env(backend="gloo" if args.share_gpu else None)
    if args.share_gpu:
        local = 0
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    ctx = _lib.get_ctx(local)
    # first-pass class capacity: 64 bits for the EEG matrices (library default 128), 32 for the audio clouds.  No window
    # of this workload needs more (tools/overflow_rate.py); any that did would be caught and repaired by the lanes.
    if not os.environ.get("TDA_CLASS_WORDS"):
        ctx.set_class_words(1, 1)

    n_win = args.windows
    wpr = args.windows_per_recording
    seg = list(range(0, n_win, wpr)) + [n_win]
    seg_off = np.array(seg, np.int32)
    n_seg = len(seg_off) - 1
    # synthetic data of the named shape; each rank its own recordings (weak scaling)
    eeg = synth.eeg_windows(n_win, seed=42 + 100000 * rank, windows_per_recording=wpr)
    aud = synth.audio_windows(n_win, args.band, seed=4242 + 100000 * rank)
    eeg_t = torch.from_numpy(eeg).to(device)
    aud_t = torch.from_numpy(aud).to(device)
    lanes = pipeline.Lanes(args.lanes, n_win, seg_off, device, graph=not args.no_graph, defer_retries=not args.no_defer)
    shards = [np.arange(r * n_seg, (r + 1) * n_seg) for r in range(world)]
    gather = (lambda res: tdist.all_gather_rows(res, shards[rank], shards, world * n_seg)) if world > 1 else None

    # dominant kernel: rips_cloud_kernel (first pass of stage rips_audio).  The one-shot probe of the C ABI is armed
    # before every launch (and before the capture of a lane's graph, so that it is baked into the replays): the kernel
    # accumulates its own duration (first workgroup start .. last workgroup end, 100 MHz wall clock) in a per-lane
    # device buffer; in the eager warm-up steps HIP events bracket the same launch on its stream.
    DOM = "rips_audio"
    span_init = np.zeros((lanes.depth, 4), np.int64)
    spans = torch.from_numpy(span_init).to(device)
    cur_events = [None]

    def arm(i):
        ev = cur_events[0]
        ctx.arm_probe(DOM, ev[0] if ev else None, ev[1] if ev else None, spans[i].data_ptr())
    lanes.before_step = arm

    def step(timers=None):
        # the inputs were uploaded before the loop and never change: no wait on the caller's stream
        return lanes.submit(eeg_t, aud_t, ctx=ctx, timers=timers, post=gather, sync_inputs=False)

    # ---- warm-up: eager steps with per-stage events (stage_ms, event_ms), then one more round of the lanes, which in
    # graph mode captures each lane's step ----
    n_eager = max(args.warmup, lanes.depth)
    ev_log, probes = [], []
    for _ in range(n_eager):
        timers = {s: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for s in pipeline.STAGES}
        cur_events[0] = (ctx.new_event(), ctx.new_event())
        probes.append(cur_events[0])
        out = step(timers)
        ev_log.append(timers)
    cur_events[0] = None
    try:
        for _ in range(lanes.depth):
            out = step()
        lanes.drain()
        torch.cuda.synchronize()
    except Exception as e:                      # graph capture refused on this stack: same steps, launched eagerly
        if not lanes.graph:
            raise
        print(f"bench: HIP graph capture failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
        torch.cuda.synchronize()
        lanes = pipeline.Lanes(args.lanes, n_win, seg_off, device, graph=False, defer_retries=not args.no_defer)
        lanes.before_step = arm
        for _ in range(lanes.depth):
            out = step()
        lanes.drain()
        torch.cuda.synchronize()
    for ws in lanes.ws:
        if not bool(((ws.eeg.status == 0) & ((ws.aud.status & ~4) == 0) & (ws.ws0 == 0) & (ws.ws1 == 0)).all()):
            raise SystemExit("bench: a window reported a non-zero status (overflow / not converged)")
    # the one HBM-streaming kernel of the step, corr_dist_kernel: HIP events around it in three steps that run alone
    hbm_ms = []
    for _ in range(3):
        evs = (ctx.new_event(), ctx.new_event())
        ctx.arm_probe("corr_dist", evs[0], evs[1])      # its own probe slot; the rips_audio probe stays armed too
        lanes.submit(eeg_t, aud_t, ctx=ctx, timers={s: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                                                   for s in pipeline.STAGES}, post=None, sync_inputs=False)
        lanes.drain()
        torch.cuda.synchronize()
        hbm_ms.append(ctx.elapsed_ms(*evs))
    stage_ms = {s: 0.0 for s in pipeline.STAGES}
    for evs in ev_log:
        for s, (a, b) in evs.items():
            stage_ms[s] += a.elapsed_time(b)
    stage_ms = {s: v / n_eager for s, v in stage_ms.items()}
    event_ms = sum(ctx.elapsed_ms(a, b) for a, b in probes) / n_eager
    sp_before = spans.cpu().numpy().copy()            # accumulators at the start of the timed region

    # ---- timed region ----
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = step()
    lanes.drain()                           # verify (and publish: all-gather at N > 1) the batches still in flight
    t_enq = time.perf_counter() - t0        # host time in the loop (it waits for the batch `lanes` steps back)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.share_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_windows = world * n_win * args.steps
        value = total_windows / dt
        dom = DOM
        sp = spans.cpu().numpy()
        launches = int(sp[:, 3].sum() - sp_before[:, 3].sum())
        assert launches == args.steps, f"probe saw {launches} launches of the dominant kernel, expected {args.steps}"
        kernel_ms = float(sp[:, 2].sum() - sp_before[:, 2].sum()) / launches / 100e6 * 1e3   # timed region only
        kernel_ms_all = float(sp[:, 2].sum()) / int(sp[:, 3].sum()) / 100e6 * 1e3            # every launch of the process
        achieved = ALG_BYTES[dom] * n_win / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        line = {
            "metric": "windows/sec end-to-end (dist->Rips H0/H1->Wasserstein), 47-ch EEG",
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: {n_win} EEG windows (47x250 f64) x 1 band ({args.band}) per GPU, "
                                   f"+ {n_win} audio windows (250 f64); corr->dist->Rips H0/H1 (EEG 47 pts, audio "
                                   f"Takens dim 3 sub 2), Wasserstein H0+H1, H1 features, per-recording "
                                   f"({wpr} windows) reductions" + ("; one all-gather of result rows" if world > 1 else ""),
                       "windows_per_gpu": n_win, "thresh": 2.0, "parallelism": f"recordings sharded x{world}",
                       "batches_in_flight": lanes.depth, "hip_graph": lanes.graph,
                       "deferred_retries": lanes.defer, "batches_repaired": lanes.repairs,
                       "first_pass_class_bits": {"eeg": 64 if not os.environ.get("TDA_CLASS_WORDS") else None, "audio": 32}},
            "stage_ms": {s: round(v, 4) for s, v in stage_ms.items()},       # eager warm-up steps, overlapping lanes
            "host_loop_ms_per_step": round(t_enq / args.steps * 1e3, 4),   # includes the wait for the batch `lanes` steps back
            "roofline": {"bound": "hbm", "kernel": "rips_cloud_kernel<512, 1, unsigned int> (stage rips_audio)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": round(kernel_ms, 4),
                         "event_ms": round(event_ms, 4), "kernel_ms_all_launches": round(kernel_ms_all, 4),
                         "launches_probed": int(sp[:, 3].sum()),
                         "alg_bytes_per_launch": ALG_BYTES[dom] * n_win,
                         "note": "irregular integer work in LDS/registers (LDS latency/issue bound); the HBM fraction is "
                                 "small by construction.  kernel_ms: average over every launch of the timed region of "
                                 "(first workgroup start .. last workgroup end), stamped by the kernel itself (100 MHz wall "
                                 "clock) -- the interval rocprofv3 --kernel-trace reports; `achieved` uses it.  "
                                 "kernel_ms_all_launches: the same over every launch of this kernel in the process (warm-up and "
                                 "capture steps included), the population rocprofv3 --stats averages.  event_ms: HIP "
                                 "events around the same launch on its stream in the eager warm-up steps; with several "
                                 "batches in flight it includes the time the grid waits for CU slots held by the others"},
        }
        cd_ms = min(hbm_ms)
        line["roofline_hbm_kernel"] = {
            "kernel": "corr_dist_kernel<3, true> (stage corr_dist)", "bound": "hbm", "achieved": ALG_BYTES["corr_dist"] * n_win / (cd_ms * 1e-3) / 1e9,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ALG_BYTES["corr_dist"] * n_win / (cd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "event_ms": round(cd_ms, 4), "alg_bytes_per_launch": ALG_BYTES["corr_dist"] * n_win,
            "traffic": (json.load(open(tpath)).get("corr_dist") if os.path.exists(tpath) else None),
            "note": "secondary: the one HBM-streaming kernel of the step (5-8 % of its GPU time); HIP events around the launch in "
                    "warm-up steps that run alone (best of 3); 710 windows fill the 256 CUs 0.9 times, the kernel reaches "
                    "2.5 TB/s on 11,360 windows (DESIGN.md 3.1)"}
        if not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(eeg, aud, seg_off, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(eeg, aud, seg_off, budget_s):
    """CPU oracle (C restatement of the ripser-class algorithm + persim's assignment), 1 core,
    timed on whole recordings of the SAME batch until ~budget_s seconds are spent."""
    from oracle import pipeline_ref      # the ONLY use of oracle/ here: the timed CPU port
    done = 0
    t0 = time.perf_counter()
    s = 0
    n_seg = len(seg_off) - 1
    while True:
        a, b = int(seg_off[s % n_seg]), int(seg_off[s % n_seg + 1])
        pipeline_ref.reference_step_cpu(eeg[a:b], aud[a:b], np.array([0, b - a]))
        done += b - a
        s += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return {"value": done / el, "unit": "windows/s", "cores": 1, "kind": "port",
            "sample": f"{done} windows ({s} recordings, cycling through the batch of {int(seg_off[-1])}), {el:.1f} s, oracle/tda_oracle.c "
                      f"(gcc -O2), same end-to-end unit"}


if __name__ == "__main__":
    main()
