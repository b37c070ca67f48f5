#!/usr/bin/env python3
"""
bench.py -- headline benchmark: windows/sec end-to-end (dist -> Rips H0/H1 -> Wasserstein), 47-channel EEG,
on N MI355X (BASELINE.json `metric`).

Workload (default `--workload corpus`, BASELINE.json configs[2] + configs[4]): the FIXED corpus of the reference --
1,416 recordings x 5 bands x 15 selected windows = 106,200 (EEG window, audio window) pairs
(scripts/tda_eeg_audio_comparison.py:39,77-80; results/eeg_audio_tda_detailed.csv has 7,080 rows of 15 windows) --
dealt over the N ranks by whole recordings (dist.shard_recordings): STRONG scaling.  A step = ONE pass of the
per-window hot path over the rank's share, resident in HBM (10 GB of float64 windows at N = 1, so nothing is served
by the 256 MiB Infinity Cache), all five bands of the share as ONE batch per pass (`--per-band`: one batch per
band), three passes in flight on their own streams as HIP graphs: per window corr->dist, Rips(EEG 47x47), tau per recording-band, Takens+Rips(audio,
23..123 points over the five bands), Wasserstein H0 and H1, H1/H0 features, the per-recording reductions, and ONE
all-gather of the (n_rec, 5 x 48) result rows per pass (RCCL over xGMI) inside the timed region -- what replaces
run_analysis' serial loop (cmp:131-138) and the partial-file merge of scripts/tda_eeg_classification_v2.py:608-638.
`--workload batch710` is BASELINE.json configs[1]: 710 windows per band-batch, five distinct batches in rotation.

A second leg (`features_pass`, BASELINE.json configs[2]) times the EEG half alone on the min-equalised 39 windows
per recording-band (276,120 windows -> the (1416, 220) feature matrix of v2:404-436,499-606, one all-gather).

`python bench.py --gpus N` with WORLD_SIZE unset starts the N ranks itself (child processes, before this process
touches a GPU); under torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE.  Rank 0 prints ONE JSON line.
`roofline` is for the dominant kernel (audio Rips), its launch duration measured live (kernel-stamped wall clock
over every launch of the timed region + HIP events on the launch stream in the eager warm-up); `roofline_lds`
restates the resource that actually binds it from the committed rocprofv3 PMC summary.  `cpu_baseline` times the
CPU oracle (oracle/tda_oracle.c rebuilt -O3 -march=native on this host; kind "port") on a bounded sample of the same
workload: one core and all the cores this process may use.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# ALGORITHMIC HBM bytes per unit of work of each kernel
ALG_BYTES = {
    "corr_dist": 47 * 250 * 8 + 47 * 47 * 8,          # window in, distance matrix out (no corr copy)
    "rips_eeg": 47 * 47 * 8 + 1100,                   # 18.7 KB/window from the stored f64 matrix
    "eeg_fused": 47 * 250 * 8 + 888 + 176,            # 95.1 KB/window: window in, diagrams + features out
    "rips_audio": 250 * 8 + 4 + 1500,                 # ~3.5 KB/window
    "wasserstein": 2700 + 8,                          # <= 2.7 KB/pair
}
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_CLOCK_HZ = 2.4e9     # MI355X_MICROARCH.md: peak engine clock


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed passes (40 x 35 ms = 1.4 s at N = 1)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="corpus", choices=["corpus", "batch710"])
    ap.add_argument("--recordings", type=int, default=1416, help="recordings of the corpus (README.md:7)")
    ap.add_argument("--windows-per-recording", type=int, default=15, help="cmp:39 MAX_WINDOWS")
    ap.add_argument("--features-windows", type=int, default=39, help="v2 min-equalised windows per recording-band")
    ap.add_argument("--features-steps", type=int, default=3, help="passes of the features leg (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=24.0, help="budget of the cpu_baseline leg (both legs)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the features / PCIe legs")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying HIP graphs")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("TDA_LANES", "3")),
                    help="batches in flight (pipeline.Lanes), each on ONE stream.  One batch per pass: three lanes are "
                         "best on the full corpus and on the share a rank of eight holds (2: -1..-4 %%, 5: -0.5..-2 %%; "
                         "tools/share_ab.sh); one batch per band (--per-band, --workload batch710): one lane per band")
    ap.add_argument("--class-words", default=os.environ.get("TDA_CLASS_WORDS", "1,1"),
                    help="first-pass class capacity (x64 bits for EEG, x32/x64 for audio); windows that need more are "
                         "redone by the widening passes inside the same step and counted in windows_repaired")
    ap.add_argument("--per-band", action="store_true", default=os.environ.get("TDA_PER_BAND", "0") == "1",
                    help="one batch per band and pass (five launches per stage) instead of all bands of the rank's share "
                         "in one batch")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 and the all-gather goes through gloo "
                         "(exercises the N>1 code path on a one-GPU box; numbers are meaningless)")
    ap.add_argument("--dump-rows", default=None, help="rank 0 saves the gathered (n_rec, 5 x 48) result rows of the last "
                                                      "pass to this .npy file (tests compare N ranks with one)")
    ap.add_argument("--cpu-worker", default=None, help=argparse.SUPPRESS)     # internal: the cpu_baseline child process
    args = ap.parse_args()
    return args


# ------------------------------------------------------------------------------------------------------------
# N ranks from a plain `python bench.py --gpus N`: children are started BEFORE this process touches a GPU
# ------------------------------------------------------------------------------------------------------------
def self_launch(args):
    import socket
    import torch                                  # device_count() does not initialise the GPU on this image
    n = args.gpus
    if not args.share_gpu and torch.cuda.device_count() < n:
        print(f"bench: --gpus {n} but only {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0:                     # one rank failed: the others would wait at a barrier for ever
                    rc = rc or code
                    for q in procs:
                        q.terminate()
            time.sleep(0.05)
    except KeyboardInterrupt:
        for q in procs:
            q.terminate()
        rc = 130
    return rc


def workload_spec(args):
    """(recording ids, windows per recording-band in the end-to-end pass, label).  batch710: 47 recordings of 15
    selected windows + one of 5 = the 710 windows of BASELINE.json configs[1] per band-batch."""
    if args.workload == "batch710":
        return 48, np.array([15] * 47 + [5]), "configs[1]"
    return args.recordings, np.full(args.recordings, args.windows_per_recording), "configs[4]+configs[2]"


def main():
    args = parse_args()
    if args.cpu_worker:
        return cpu_worker(args.cpu_worker)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)

    import torch
    import torch.distributed as dist
    from tda_eeg_audio_amd import _lib, pipeline, synth
    from tda_eeg_audio_amd import dist as tdist

    rank, world, local = tdist.init_from_env(backend="gloo" if args.share_gpu else None)
    if world != args.gpus:
        print(f"bench: WORLD_SIZE={world} but --gpus {args.gpus}", file=sys.stderr)
        return 2
    if args.share_gpu:
        local = 0
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X: the HIP path has no CPU fallback", file=sys.stderr)
        return 2
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    ctx = _lib.get_ctx(local)
    cw_dm, cw_cloud = (int(x) for x in args.class_words.split(","))
    ctx.set_class_words(cw_dm, cw_cloud)
    retry_ctr = torch.zeros(4, dtype=torch.int64, device=device)
    ctx.set_retry_counter(retry_ctr.data_ptr())

    # ---- the fixed workload and this rank's share of it (whole recordings; strong scaling) ----
    n_rec, counts, cfg_label = workload_spec(args)
    uniform = bool((counts == counts[0]).all())
    wpr = int(counts[0])
    shards = tdist.shard_recordings(counts, world)
    mine = shards[rank]
    bands = synth.BANDS
    nb = len(bands)
    t_gen = time.perf_counter()
    if uniform:
        eeg = synth.corpus_eeg_dev(mine, wpr, nb, device, seed=42)
        aud_all = synth.corpus_audio(n_rec, wpr, bands, seed=4242)
        aud = [torch.from_numpy(np.ascontiguousarray(aud_all[b][mine].reshape(-1, 250))).to(device) for b in bands]
        runner = pipeline.CorpusPass(eeg, aud, wpr, device, ctx, depth=args.lanes, graph=not args.no_graph,
                                     my_recs=mine, shards=shards, n_total=n_rec, merge_bands=not args.per_band)
    else:                                         # batch710: ragged last recording -> explicit group offsets
        assert world == 1, "--workload batch710 is the one-GPU configuration"
        seg_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        eeg15 = synth.corpus_eeg_dev(np.arange(n_rec), 15, nb, device, seed=42)
        aud_all = synth.corpus_audio(n_rec, 15, bands, seed=4242)
        keep = np.concatenate([np.arange(r * 15, r * 15 + counts[r]) for r in range(n_rec)])
        keep_t = torch.from_numpy(keep).to(device)
        eeg = [e[keep_t].contiguous() for e in eeg15]
        aud = [torch.from_numpy(np.ascontiguousarray(aud_all[b].reshape(-1, 250)[keep])).to(device) for b in bands]
        del eeg15
        runner = pipeline.CorpusPass(eeg, aud, None, device, ctx, depth=args.lanes, graph=not args.no_graph,
                                     seg_off=seg_off)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t_gen
    n_win_batch = runner.n_win                    # windows per batch (= per launch of every stage) on this rank
    n_batches = len(runner.batches)
    total_pairs = int(counts.sum()) * nb          # window pairs of one pass of the whole job
    lanes = runner.lanes

    # dominant kernel: rips_cloud_kernel (first pass of stage rips_audio).  The one-shot probe of the C ABI is armed
    # before every eager launch and before the capture of a band's graph, so that it is baked into the replays: the
    # kernel accumulates its own duration (first workgroup start .. last workgroup end, 100 MHz wall clock) in a
    # per-lane device buffer; in the eager warm-up pass HIP events bracket the same launch on its stream.
    DOM = "rips_audio"
    spans = torch.zeros((lanes.depth, 4), dtype=torch.int64, device=device)
    cur_events = [None]

    def arm(i):
        ev = cur_events[0]
        ctx.arm_probe(DOM, ev[0] if ev else None, ev[1] if ev else None, spans[i].data_ptr())
    lanes.before_step = arm

    # ---- warm-up: one eager pass with per-stage events (stage_ms, event_ms), then W passes (the first captures) ----
    ev_log, probes = [], []
    for b, (eb, ab, _) in enumerate(runner.batches):
        timers = {s: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for s in pipeline.STAGES}
        cur_events[0] = (ctx.new_event(), ctx.new_event())
        probes.append(cur_events[0])
        lanes.submit(eb, ab, ctx=ctx, timers=timers, sync_inputs=False, lane=b)
        lanes.drain()
        ev_log.append(timers)
    cur_events[0] = None
    torch.cuda.synchronize()
    def _ms(t, s):
        try:
            return t[s][0].elapsed_time(t[s][1])
        except Exception:                        # stage not on this path (fused / two-kernel EEG chain)
            return 0.0
    stage_ms = {s: sum(_ms(t, s) for t in ev_log) for s in pipeline.STAGES}    # per pass, every stage alone on the GPU
    stage_ms = {s: v for s, v in stage_ms.items() if v > 0.0}
    event_ms = [ctx.elapsed_ms(a, b) for a, b in probes]
    try:
        runner.prime()                          # every lane captures its graph (set-up, not a warm-up step)
        for _ in range(max(1, args.warmup)):
            runner.step()
        rows = runner.finish()
        torch.cuda.synchronize()
    except Exception as e:                      # graph capture refused on this stack: same steps, launched eagerly
        if not lanes.graph:
            raise
        print(f"bench: HIP graph capture failed ({type(e).__name__}: {e}); falling back to eager launches", file=sys.stderr)
        torch.cuda.synchronize()
        lanes.graph = False
        lanes.graphs.clear()
        for _ in range(max(1, args.warmup)):
            runner.step()
        rows = runner.finish()
        torch.cuda.synchronize()
    bad = 0
    for ws in lanes.ws:
        ok = (ws.eeg.status == 0) & ((ws.aud.status & ~4) == 0) & (ws.ws0 == 0) & (ws.ws1 == 0)
        bad += int((~ok).sum().item())
    if bad:
        print(f"bench: {bad} windows reported a non-zero status (overflow beyond the ladder / not converged)", file=sys.stderr)
        return 3
    sp_before = spans.cpu().numpy().copy()            # accumulators at the start of the timed region
    retry_before = retry_ctr.cpu().numpy().copy()

    # ---- timed region ----
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        runner.step()
    rows = runner.finish()                  # verify and publish (all-gather at N > 1) the batches still in flight
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    dt_rank = [dt]
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device="cpu" if args.share_gpu else device)
        t[rank] = dt
        dist.all_reduce(t, op=dist.ReduceOp.SUM)          # every rank's own time; the job's time is the slowest rank's
        dt_rank = [float(x) for x in t.tolist()]
        dt = max(dt_rank)
    if args.dump_rows and rank == 0:
        np.save(args.dump_rows, rows.cpu().numpy())
    sp = spans.cpu().numpy()
    retry_after = retry_ctr.cpu().numpy()
    assert rows.shape == ((n_rec if world > 1 else len(mine)), nb * pipeline.RESULT_COLS)
    finite_frac = float(torch.isfinite(rows).double().mean().item())

    # ---- secondary legs (outside the headline's timed region) ----
    extras = {}
    if not args.no_extras and uniform:
        if args.features_steps > 0:
            extras["features_pass"] = features_leg(args, ctx, device, mine, shards, n_rec, rank, world, nb)
        if rank == 0:
            extras["pcie_inclusive"] = pcie_leg(ctx, device, eeg[nb - 2], aud[nb - 2], wpr)
            extras["front_end_inclusive"] = front_end_leg(ctx, device, args.features_windows)
            extras["recordings_from_host"] = recordings_leg(ctx, device)

    if rank == 0:
        value = total_pairs * args.steps / dt
        launches = int(sp[:, 3].sum() - sp_before[:, 3].sum())
        exp = args.steps * n_batches
        assert launches == exp, f"probe saw {launches} launches of the dominant kernel, expected {exp}"
        kernel_ms = float(sp[:, 2].sum() - sp_before[:, 2].sum()) / launches / 100e6 * 1e3   # timed region only
        kernel_ms_all = float(sp[:, 2].sum()) / int(sp[:, 3].sum()) / 100e6 * 1e3            # every launch of the process
        achieved = ALG_BYTES[DOM] * n_win_batch / (kernel_ms * 1e-3) / 1e9

        def prof(name):
            path = os.path.join(ROOT, "profiles", name)
            try:
                return json.load(open(path))
            except Exception:
                return None
        tr = prof("traffic.json") or {}
        traffic = tr.get(DOM)
        if traffic is not None and tr.get("_windows_per_launch"):        # counted at another launch size: scale it
            traffic = int(round(traffic * n_win_batch / tr["_windows_per_launch"]))
        line = {
            "metric": "windows/sec end-to-end (dist->Rips H0/H1->Wasserstein), 47-ch EEG",
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": max(1, args.warmup),
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.workload == "corpus" else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{cfg_label}: {n_rec} recordings x {nb} bands x {wpr if uniform else '15 (last: 5)'} "
                                   f"selected windows = {total_pairs} (EEG 47x250 f64, audio 250 f64) window pairs per "
                                   f"pass, the whole job; per pair corr->dist->Rips H0/H1 (EEG 47 pts), tau per "
                                   f"recording-band, Takens dim 3 sub 2 + Rips (audio, 23..123 pts over the bands), "
                                   f"Wasserstein H0+H1, features, per-recording reductions; one all-gather of the "
                                   f"({n_rec}, {nb}x48) rows per pass",
                       "windows_per_pass": total_pairs, "windows_per_gpu_per_pass": int(n_win_batch * n_batches),
                       "input_bytes_per_gpu": int(sum(e.numel() for e in eeg) * 8 + sum(a.numel() for a in aud) * 8),
                       "batches_per_pass": n_batches, "windows_per_batch": int(n_win_batch), "thresh": 2.0,
                       "parallelism": f"recordings dealt over {world} rank(s), one all-gather per pass",
                       "batches_in_flight": lanes.depth, "hip_graph": lanes.graph,
                       "ms_per_step_per_rank": [round(x / args.steps * 1e3, 4) for x in dt_rank],
                       "allgather_ms": runner.allgather_ms(),
                       "first_pass_class_bits": {"eeg": 64 * cw_dm, "audio": 32 * cw_cloud if cw_cloud == 1 else 64},
                       "windows_repaired": {"eeg": int(retry_after[0] - retry_before[0]),
                                            "audio": int(retry_after[1] - retry_before[1]),
                                            "last_rung": int(retry_after[2] - retry_before[2]),
                                            "note": "windows redone by the widening passes inside the timed steps"},
                       "batches_rerun_with_full_ladder": lanes.repairs,
                       "result_rows_finite_frac": round(finite_frac, 6),
                       "data_generation_s": round(t_gen, 2)},
            "stage_ms": {s: round(v, 4) for s, v in stage_ms.items()},     # one eager pass, every stage alone on the GPU
            "host_loop_ms_per_step": round(t_enq / args.steps * 1e3, 4),
            "roofline": {"bound": "hbm", "kernel": "rips_cloud_kernel<512, 1, unsigned int, false, true> (stage rips_audio: first pass, "
                                                   "narrow layout, three workgroups per CU)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": round(kernel_ms, 4),
                         "event_ms": round(float(np.mean(event_ms)), 4), "event_ms_per_batch": [round(x, 4) for x in event_ms],
                         "kernel_ms_all_launches": round(kernel_ms_all, 4),
                         "launches_probed": int(sp[:, 3].sum()),
                         "alg_bytes_per_launch": ALG_BYTES[DOM] * n_win_batch,
                         "note": "irregular integer work in LDS/registers: the resource that binds it is LDS latency / "
                                 "issue (see roofline_lds), so the HBM fraction is small by construction.  kernel_ms: "
                                 "average over every launch of the timed region of (first workgroup start .. last "
                                 "workgroup end), stamped by the kernel itself (100 MHz wall clock) -- the interval "
                                 "rocprofv3 --kernel-trace reports; `achieved` uses it.  event_ms: HIP events around the "
                                 "same launch on its stream, alone on the GPU (eager warm-up pass)"},
            "roofline_lds": prof("r03_lds_roofline.json") or prof("r02_lds_roofline.json"),
        }
        vp = (line["roofline_lds"] or {}).get("valu_per_window")
        if vp and args.workload == "corpus":
            # the step as a whole against the vector-issue roof: a SIMD-32 issues a wave64 instruction over 2 cycles
            # once two or more waves share it (MI355X_MICROARCH.md, row v_fma_f32: "2 cyc (SIMD-32); one wave alone: 4")
            peak = world * 256 * 4 * VALU_CLOCK_HZ / 2.0
            ach = vp["total"] * value
            line["roofline_valu"] = {"bound": "valu_issue", "achieved": ach / 1e9, "peak": peak / 1e9,
                                     "unit": "G wave-instructions/s", "frac": ach / peak,
                                     "valu_wave_instructions_per_window": vp["total"],
                                     "note": "instructions per window pair from the committed SQ_INSTS_VALU counters "
                                             "(profiles/r03_sq_counters.json, every kernel of the step), rate from this "
                                             "run; peak = CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 instruction (SIMD-32, two or "
                                             "more waves per SIMD: the guide's figure).  frac_of_measured_issue_rate: against "
                                             "4 cycles per instruction at the 2.13 GHz the chip holds under load -- what "
                                             "tools/probes/valu_rate.hip measures for the integer / 64-bit / f64 instructions "
                                             "these kernels consist of (profiles/r03_valu_rate.txt; only v_add_u32, v_and_b32 "
                                             "and f32 fma issue in 2)",
                                     "frac_of_measured_issue_rate": ach / (world * 256 * 4 * 2.13e9 / 4.0)}
        fp = extras.get("features_pass")
        if fp:                                   # secondary: the one HBM-streaming kernel of the step
            nw, ms = fp.pop("eeg_kernel_windows"), fp.pop("eeg_kernel_ms")
            ach = ALG_BYTES["eeg_fused"] * nw / (ms * 1e-3) / 1e9
            tre = tr.get("eeg_window")
            line["roofline_hbm_kernel"] = {
                "kernel": "eeg_window_kernel<3, false, 1, false> (fused EEG window: samples -> corr -> dist -> Rips)",
                "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "event_ms": round(ms, 4), "windows_per_launch": nw, "alg_bytes_per_launch": ALG_BYTES["eeg_fused"] * nw,
                "traffic": int(round(tre * nw / tr["_windows_per_launch"])) if tre and tr.get("_windows_per_launch") else None,
                "note": "secondary: the kernel that streams the corpus (95.1 KB per window, SURVEY.md section 8d); HIP "
                        "events around one launch of the EEG-only features leg, nothing else on the GPU.  Its time is "
                        "the Rips sweep behind the fetch, not the fetch.  traffic: the window is fetched twice (means, then "
                        "products); the second fetch misses L2 and is counted by FETCH_SIZE, but the windows in flight (96 MB) "
                        "sit in the 256 MiB Infinity Cache, which the counter does not tell from HBM; TDA_EEG_RESIDENT=1 "
                        "keeps the window in registers instead (one fetch, 11 % slower on this leg)"}
        line.update(extras)
        if not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline(eeg, aud, wpr if uniform else 15, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def features_leg(args, ctx, device, mine, shards, n_rec, rank, world, nb):
    """BASELINE.json configs[2]: EEG Rips + features on the min-equalised windows of every recording-band
    (v2:79-82,519-527: 39), (n_rec, 220) matrix all-gathered once per pass."""
    import torch
    import torch.distributed as dist
    from tda_eeg_audio_amd import pipeline, synth
    from tda_eeg_audio_amd import dist as tdist
    fw = args.features_windows
    eeg = synth.corpus_eeg_dev(mine, fw, nb, device, seed=4343)
    n_win = eeg[0].shape[0]
    seg_off = np.arange(0, n_win + 1, fw, dtype=np.int32)
    ws = pipeline.Workspace(n_win, seg_off, device)
    block = torch.empty((len(mine), nb, 44), dtype=torch.float64, device=device)

    ctx.set_retry_policy(ctx.RETRY_ONE_STEP)      # (a status word left non-zero shows in windows_bad_status below)

    def one_pass():
        for b in range(nb):
            block[:, b].copy_(pipeline.run_features_step(eeg[b], ws, ctx=ctx))
        flat = block.view(len(mine), nb * 44)
        return tdist.all_gather_rows(flat, mine, shards, n_rec) if world > 1 else flat
    X = one_pass()
    torch.cuda.synchronize()
    # the fused EEG kernel alone (nothing else on the GPU): HIP events around its first-pass launch on its stream
    evs = (ctx.new_event(), ctx.new_event())
    ctx.arm_probe("rips_eeg", evs[0], evs[1])
    pipeline.run_features_step(eeg[nb - 2], ws, ctx=ctx)
    torch.cuda.synchronize()
    eeg_kernel_ms = ctx.elapsed_ms(*evs)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.features_steps):
        X = one_pass()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.share_gpu else device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ctx.set_retry_policy(ctx.RETRY_AUTO)
    bad = int((ws.eeg.status != 0).sum().item())
    total = n_rec * nb * fw
    del eeg, ws
    torch.cuda.empty_cache()
    return {"workload": f"configs[2]: {n_rec} recordings x {nb} bands x {fw} windows = {total} EEG windows -> "
                        f"({n_rec}, {nb * 44}) feature matrix, one all-gather per pass",
            "value": total * args.features_steps / dt, "unit": "windows/s", "steps": args.features_steps,
            "ms_per_step": dt / args.features_steps * 1e3, "matrix_shape": list(X.shape),
            "matrix_finite": bool(torch.isfinite(X).all().item()), "windows_bad_status": bad,
            "eeg_kernel_ms": eeg_kernel_ms, "eeg_kernel_windows": int(n_win)}


def pcie_leg(ctx, device, eeg_b, aud_b, wpr, n_rec=1024):
    """PCIe-inclusive rate (never `value`): pinned host windows -> HBM -> the same step -> result rows back to the
    host, copies and kernels on one stream; a bounded sample of one band's batch.  (One stream on purpose: with the
    upload of the next batch on its own stream beside the step the copy drops from 57 to 42 GB/s and the leg from 0.47 to
    0.44 M windows/s -- measured in round 3; the leg is bound by the link either way: 94 KB per window.)"""
    import torch
    from tda_eeg_audio_amd import pipeline
    n_win = min(eeg_b.shape[0], n_rec * wpr)
    n_win -= n_win % wpr
    eeg_h = eeg_b[:n_win].cpu().pin_memory()
    aud_h = aud_b[:n_win].cpu().pin_memory()
    eeg_d = torch.empty_like(eeg_h, device=device)
    aud_d = torch.empty_like(aud_h, device=device)
    seg_off = np.arange(0, n_win + 1, wpr, dtype=np.int32)
    ws = pipeline.Workspace(n_win, seg_off, device)
    out_h = torch.empty((len(seg_off) - 1, pipeline.RESULT_COLS), dtype=torch.float64).pin_memory()

    def step():
        eeg_d.copy_(eeg_h, non_blocking=True)
        aud_d.copy_(aud_h, non_blocking=True)
        out_h.copy_(pipeline.run_step(eeg_d, aud_d, ws, ctx=ctx), non_blocking=True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 8
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    nbytes = eeg_h.numel() * 8 + aud_h.numel() * 8
    return {"value": n_win / dt, "unit": "windows/s", "stage": "h2d + step + d2h, one stream, pinned host buffers",
            "sample": f"{n_win} window pairs of one band ({nbytes / 1e6:.0f} MB uploaded per step)",
            "h2d_GBps": nbytes / dt / 1e9, "ms_per_step": dt * 1e3}


def front_end_leg(ctx, device, n_sel, n_rec=354, n_samples=4606):
    """Front-end-inclusive rate of the EEG half (never `value`): RAW recordings (47 x 4,606 samples = 71 windows, the
    first row of results/preprocessing_metadata.csv) in pinned host memory -> HBM -> five zero-phase band-passes of all
    channels -> fused window kernel on the sliding windows read in place (n_sel evenly spaced windows per
    recording-band) -> features -> (n_rec, 220) matrix back on the host.  1.73 MB uploaded per recording instead of the
    33 MB of its five (71, 47, 250) window stacks (43 GB for the corpus)."""
    import torch
    from tda_eeg_audio_amd import preprocess
    g = torch.Generator(device=device)
    g.manual_seed(777)
    raw_d = torch.randn((n_rec, 47, n_samples), generator=g, dtype=torch.float64, device=device)
    raw_d += 0.5 * torch.randn((n_rec, 1, n_samples), generator=g, dtype=torch.float64, device=device)
    raw_h = raw_d.cpu().pin_memory()
    per_rec = (n_samples - 250) // 62 + 1
    pick = np.linspace(0, per_rec - 1, n_sel, dtype=int)
    sel = (np.arange(n_rec)[:, None] * per_rec + pick[None, :]).astype(np.int32).ravel()
    sel_t = torch.from_numpy(sel).to(device)
    out_h = torch.empty((n_rec, 220), dtype=torch.float64).pin_memory()

    def step():
        raw_d.copy_(raw_h, non_blocking=True)
        X, st = preprocess.recordings_to_features(raw_d, 250, sel_t=sel_t, n_sel_per_rec=n_sel, ctx=ctx)
        out_h.copy_(X, non_blocking=True)
        return st
    st = step()
    torch.cuda.synchronize()
    bad = int((st != 0).sum().item())
    t0 = time.perf_counter()
    K = 4
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    n_win = n_rec * 5 * n_sel
    return {"value": n_win / dt, "unit": "windows/s", "ms_per_step": dt * 1e3,
            "stage": "raw EEG h2d + 5 x sosfiltfilt (all channels, one launch each) + fused sliding-window kernel + features "
                     "+ aggregation + d2h, one stream",
            "sample": f"{n_rec} recordings x 47 channels x {n_samples} samples ({raw_h.numel() * 8 / 1e6:.0f} MB uploaded per step), "
                      f"{n_sel} of {per_rec} windows per recording-band = {n_win} windows",
            "matrix_finite": bool(torch.isfinite(out_h).all().item()), "windows_bad_status": bad}


def recordings_leg(ctx, device, n_rec=1416, shard=236, n_samples=4606):
    """process_recording whole, from HOST memory (never `value`): raw EEG (47 x 4,606 float64 = 71 windows) and the 250 Hz
    audio envelope of every recording in pinned host buffers -> per shard: upload, five zero-phase band-passes of both,
    the 15 selected windows per recording-band read in place -> corr -> dist -> Rips | tau -> Takens -> Rips ->
    Wasserstein -> rows -> back to the host; the upload of shard k + 1 overlaps the compute of shard k
    (recordings.RecordingPass).  The 1,416 recordings of the corpus per run, twice."""
    import torch
    from tda_eeg_audio_amd import recordings
    g = torch.Generator(device="cpu")
    g.manual_seed(909)
    raw_h = torch.randn((n_rec, 47, n_samples), generator=g, dtype=torch.float64)
    raw_h += 0.5 * torch.randn((n_rec, 1, n_samples), generator=g, dtype=torch.float64)
    raw_h = raw_h.pin_memory()
    env_h = (torch.randn((n_rec, n_samples), generator=g, dtype=torch.float64).abs()
             + 0.3 * torch.randn((n_rec, n_samples), generator=g, dtype=torch.float64).cumsum(1).abs() * 0.02).pin_memory()
    rp = recordings.RecordingPass(n_samples, shard, device, ctx=ctx)
    rows = rp.run(raw_h, env_h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 2
    for _ in range(K):
        rows = rp.run(raw_h, env_h, rows)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    n_pairs = n_rec * 5 * rp.k
    nbytes = raw_h.numel() * 8 + env_h.numel() * 8
    return {"value": n_pairs / dt, "unit": "window pairs/s", "ms_per_run": dt * 1e3, "h2d_GBps": nbytes / dt / 1e9,
            "stage": "pinned raw EEG + 250 Hz envelope -> h2d -> 5 x (sosfiltfilt of all channels + filtfilt of the envelopes) "
                     "-> fused EEG window kernel on sliding windows + tau / Takens / Rips on the selected envelope windows "
                     "-> finish -> Wasserstein H0 / H1 -> rows -> d2h; shards double-buffered, one stream per band",
            "sample": f"{n_rec} recordings x 47 channels x {n_samples} samples + envelopes ({nbytes / 1e6:.0f} MB uploaded per "
                      f"run) in shards of {shard}; {rp.k} of {rp.per_rec} windows per recording-band = {n_pairs} window pairs",
            "rows_finite": bool(torch.isfinite(rows).all().item()), "band_steps_rerun_with_full_ladder": rp.repairs}


# ------------------------------------------------------------------------------------------------------------
# cpu_baseline: the CPU oracle on a bounded sample of the same workload, in a child process that never touches a GPU
# ------------------------------------------------------------------------------------------------------------
def cpu_baseline(eeg, aud, wpr, budget_s, n_rec=8):
    """Sample = the first recordings of this rank's share, all five bands.  The child process (this file with
    --cpu-worker) rebuilds oracle/tda_oracle.c with -O3 -march=native on THIS host and times orc_segment_step
    (one (recording, band) group end to end, in C): first on one core, then on all the cores it may use with a
    process pool over the groups (the reference's joblib processes, v2:569-572)."""
    import tempfile
    n = min(n_rec, eeg[0].shape[0] // wpr) * wpr
    tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=tmpdir) as td:
        path = os.path.join(td, "sample.npz")
        np.savez(path, eeg=np.stack([e[:n].cpu().numpy() for e in eeg]), aud=np.stack([a[:n].cpu().numpy() for a in aud]),
                 wpr=wpr, budget=budget_s)
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-worker", path], env=env,
                             capture_output=True, text=True, timeout=max(300.0, 20 * budget_s))
    if out.returncode != 0:
        return {"error": out.stderr[-500:]}
    return json.loads(out.stdout.strip().splitlines()[-1])


_CPU_GROUPS = None


def _cpu_group(i):
    from oracle import port
    e, a = _CPU_GROUPS[i % len(_CPU_GROUPS)]
    t0 = time.perf_counter()
    port.segment_step(e, a)
    return time.perf_counter() - t0


def _cpu_load(path):
    """(eeg, aud) of every (recording, band) group of the sample, recording-major."""
    global _CPU_GROUPS
    z = np.load(path)
    eeg, aud, wpr = z["eeg"], z["aud"], int(z["wpr"])
    nb, n = eeg.shape[0], eeg.shape[1]
    _CPU_GROUPS = [(np.ascontiguousarray(eeg[b, r:r + wpr]), np.ascontiguousarray(aud[b, r:r + wpr]))
                   for r in range(0, n, wpr) for b in range(nb)]
    return wpr, nb, n, float(z["budget"])


def _cpu_init(path):
    from oracle import port
    port.use_native()
    _cpu_load(path)                         # every worker holds the sample: a job is an index, nothing is pickled


def cpu_worker(path):
    import multiprocessing as mp
    from oracle import port                 # the ONLY use of oracle/ in this file: the timed CPU port
    flags = port.use_native()
    wpr, nb, n, budget = _cpu_load(path)
    cpu_model = "unknown"
    phys = set()
    try:
        pid = cid = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                cid = ln.split(":")[1].strip()
                phys.add((pid, cid))
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0))
    # all PHYSICAL cores of the host (the reference's joblib pool, v2:569-572), unless told
    # otherwise; the rate on 16 workers (one GPU's share of the host) is reported beside it
    n_phys = len(phys) or avail
    workers = max(1, min(avail, int(os.environ.get("TDA_CPU_WORKERS", str(n_phys)))))
    # ---- one core: >= 5 samples, each one recording (nb groups) ----
    _cpu_group(0)                                                    # warm
    samples1, t_spent, i = [], 0.0, 0
    while len(samples1) < 5 or (t_spent < 0.4 * budget and len(samples1) < 50):
        t = sum(_cpu_group(i * nb + j) for j in range(nb))
        samples1.append(nb * wpr / t)
        t_spent += t
        i += 1
    # ---- all cores: a pool over the groups, >= 5 samples of `workers` x 4 groups each ----
    def pool_rate(nw, budget_s):
        out = []
        with mp.get_context("fork").Pool(nw, initializer=_cpu_init, initargs=(path,)) as pool:
            batch = list(range(4 * nw))
            pool.map(_cpu_group, batch, chunksize=1)                     # warm
            t_spent = 0.0
            while len(out) < 5 or (t_spent < budget_s and len(out) < 50):
                t0 = time.perf_counter()
                pool.map(_cpu_group, batch, chunksize=1)
                t = time.perf_counter() - t0
                out.append(len(batch) * wpr / t)
                t_spent += t
        return out
    samples_n = pool_rate(workers, 0.4 * budget)
    samples_16 = pool_rate(16, 0.15 * budget) if workers != 16 and avail >= 16 else None
    v1, vn = float(np.median(samples1)), float(np.median(samples_n))
    print(json.dumps({
        "value": v1, "unit": "windows/s", "cores": 1, "kind": "port",
        "value_all_cores": vn, "cores_all": workers,
        "value_16_workers": float(np.median(samples_16)) if samples_16 else None, "cores_total": os.cpu_count(), "cores_physical": len(phys) or None,
        "cores_available": avail, "cpu_model": cpu_model, "build": flags,
        "sample": f"oracle/tda_oracle.c::orc_segment_step ({flags}), the same end-to-end unit on whole (recording, band) "
                  f"groups of {wpr} window pairs drawn from {n // wpr} recordings x {nb} bands of this workload; 1 core: "
                  f"median of {len(samples1)} samples of {nb} groups; all cores: process pool of {workers} over the "
                  f"groups (v2:569-572 joblib processes), median of {len(samples_n)} samples of {4 * workers} groups (value_16_workers: the same on 16); "
                  f"RESTATED baseline (ripser/persim are not installable offline)",
        "samples_1core": [round(x, 1) for x in samples1[:10]], "samples_all_cores": [round(x, 1) for x in samples_n[:10]]}))
    return 0


if __name__ == "__main__":
    sys.exit(main() or 0)
