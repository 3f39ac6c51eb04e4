"""profiles/rNN_final_bench_stats.md (+ rNN_bench_line.json) from a rocprofv3 --kernel-trace --stats run of the default
bench command (tools/prof.sh final ...):
    python tools/make_final_profile.py gpurun_out/prof_final gpurun_out/prof_final.log [round = 2] [unprofiled bench json]"""
import csv, json, os, sys

d, log = sys.argv[1], sys.argv[2]
rnd = int(sys.argv[3]) if len(sys.argv) > 3 else 2
unprof = json.loads([l for l in open(sys.argv[4]) if l.startswith('{"metric"')][-1]) if len(sys.argv) > 4 else None
line = [l for l in open(log) if l.startswith('{"metric"')][-1]
bench = json.loads(line)
stats = list(csv.DictReader(open(os.path.join(d, "bench_kernel_stats.csv"))))
trace = list(csv.DictReader(open(os.path.join(d, "bench_kernel_trace.csv"))))
out = []
out.append(f"# Round {rnd} — final state: rocprofv3 `--kernel-trace --stats` of the DEFAULT bench command\n")
out.append("Command (MI355X box): `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o bench -- "
           "python bench.py` (`tools/prof.sh final`; default flags `--gpus 1 --steps 200 --warmup 20`, all legs: timed PPO "
           "workload, kernel breakdown, env-step roofline at 2^24 envs, roofline extras incl. the DQN / MFMA / replay "
           "configs, CPU baseline).\n")
out.append("Bench line printed by the same (profiled) run -- the profiler intercepts every launch, so `ms_per_step` / `value` of a "
           "profiled run can sit above the unprofiled ones"
           + (f" ({unprof['ms_per_step']} ms, {unprof['value']:.3e} env-steps/s for this code, `profiles/r{rnd:02d}_bench_line.json`)" if unprof else "")
           + "; the per-kernel durations below are what this file is for:\n\n```json\n" + json.dumps(bench, indent=1) + "\n```\n")
out.append("| kernel | calls | total us | avg us | % |\n|---|---|---|---|---|")
for r in stats:
    name = r["Name"].split("(")[0].replace("void ", "")
    if "rlhip" not in name:
        continue
    out.append(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e3:.1f} | {float(r['AverageNs'])/1e3:.2f} | "
               f"{float(r['Percentage']):.2f} |")
# agreement check for the roofline kernel: the 2^24-env launches only (grid 16384 x 256 threads = 4194304 work-items)
big = [float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in trace
       if "env_step_kernel" in r["Kernel_Name"] and "CartPole" in r["Kernel_Name"] and int(r["Grid_Size_X"]) >= (1 << 22)]
rf = bench.get("roofline", {})
if big:
    # the roofline leg: 3 warm-up launches, forced reset, 1 + 5 launches before any env can terminate (the 5 are timed),
    # 60 launches that de-synchronise the episodes, then the 20 timed steady-state launches
    nt, desync, timed = big[4:9], big[9:69], big[69:89]
    us = lambda v: f"{sum(v)/len(v)/1e3:.1f}"
    out.append("")
    out.append(f"**Agreement check (roofline kernel).** The stats rows of `env_step_kernel<CartPole, float, ...>` mix launch "
               f"sizes (the roofline leg runs 2^24 envs, the DQN legs 4096).  From `bench_kernel_trace.csv` ({len(big)} "
               f"launches with 2^24 envs, grid 16384 x 256): the 20 timed steady-state launches take **{us(timed)} us** on "
               f"average (min {min(timed)/1e3:.1f}, max {max(timed)/1e3:.1f}) against **{rf.get('us_per_launch')} us** "
               f"measured with HIP events inside `bench.py` in the same run -> roofline.achieved = {rf.get('achieved')} "
               f"GB/s, frac = {rf.get('frac')}; the 5 launches before any env can terminate: **{us(nt)} us** against "
               f"{rf.get('without_terminations', {}).get('us_per_launch')} us (`roofline.without_terminations`).")
    out.append("")
    out.append("Per-launch durations (us) of the 60 de-synchronising launches in between -- the termination waves of the "
               "synchronised start are visible (no env can terminate before step ~8, then bursts that flatten out): "
               + ", ".join(f"{x/1e3:.0f}" for x in desync))
open(f"profiles/r{rnd:02d}_final_bench_stats.md", "w").write("\n".join(out) + "\n")
if unprof:
    json.dump(unprof, open(f"profiles/r{rnd:02d}_bench_line.json", "w"), indent=1)
print("\n".join(out[-3:])[:1500])
