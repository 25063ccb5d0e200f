"""profiles/<round>_pmc.json (round = $ROUND, default r03) from the rocprofv3 --pmc passes of scripts/profile_pmc.sh (gpurun_out/pmc_<TAG>/summary.txt).

usage: make_profile_json.py TAG KEY LAUNCHES_PER_RUN
  KEY               bench.py's profile key, e.g. book1_1920x1080_spp512_f64
  LAUNCHES_PER_RUN  full renders per bench.py run in the pass (profile_pmc.sh: 1 counted launch + 1 step = 2)

Derived quantities (per launch = one full render; a render that runs in several sample batches dispatches the kernel
several times, their counters are summed):
  shader_cycles              GRBM_GUI_ACTIVE / 8 XCDs
  valu_instr_per_simd_cycle  SQ_INSTS_VALU / (shader_cycles * 1024 SIMDs)
  issue_busy_vs_guide_peak   that / 0.5  (MI355X guide: one wave64 v_fma_f32 per 2 cycles per SIMD; f64 instructions are
                             half rate, so for the f64 kernels this understates pipe occupancy)
  lane_utilisation           SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)
  valu_lane_roofline_frac    issue_busy_vs_guide_peak * lane_utilisation
  wave_time_split            issuing = SQ_ACTIVE_INST_ANY, waiting_on_counters = SQ_WAIT_ANY, issue_stalled = the rest, per SQ_WAVE_CYCLES
  hbm_bytes                  (FETCH_SIZE + WRITE_SIZE) KiB * 1024; FETCH_SIZE is a lower bound on gfx950 (guide, HBM section)
  l2_hit_rate                TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import json, os, sys

tag, key, per_run = sys.argv[1], sys.argv[2], float(sys.argv[3])
R = os.environ.get("ROUND", "r03")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
text = open(os.path.join(root, "gpurun_out", "pmc_" + tag, "summary.txt")).read()
sums, meta = {}, {}
for line in text.splitlines():
    if " dispatch " not in line:
        continue
    v = json.loads(line[line.index("{"):])
    for k, x in v.items():
        if k.startswith("_"):
            meta[k] = x
        else:
            sums[k] = sums.get(k, 0.0) + x
c = {k: v / per_run for k, v in sums.items()}
cycles = c["GRBM_GUI_ACTIVE"] / 8.0
valu = c["SQ_INSTS_VALU"] / (cycles * 1024.0)
lane = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0)
wc = c["SQ_WAVE_CYCLES"]
issuing, waiting = c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc
rec = {
    "kernel": meta.get("_kernel"), "workgroup_size": meta.get("_Workgroup_Size"), "rocprof_arch_vgpr_count": meta.get("_VGPR_Count"), "rocprof_arch_vgpr_count_note": "rocprofv3 reports HALF the allocation on gfx950; the allocation is in profiles/r03_kernel_resources.json",
    "shader_cycles": round(cycles), "valu_instr_per_simd_cycle": round(valu, 4), "guide_peak_instr_per_simd_cycle": 0.5,
    "issue_busy_vs_guide_peak": round(valu / 0.5, 4), "lane_utilisation": round(lane, 4),
    "valu_lane_roofline_frac": round(valu / 0.5 * lane, 4),
    "valu_wave_instr": round(c["SQ_INSTS_VALU"]), "salu_wave_instr": round(c["SQ_INSTS_SALU"]), "lds_wave_instr": round(c["SQ_INSTS_LDS"]),
    "vmem_wave_instr": round(c["SQ_INSTS_VMEM"]),
    "wave_time_split": {"issuing": round(issuing, 3), "waiting_on_counters": round(waiting, 3), "issue_stalled": round(max(0.0, 1 - issuing - waiting), 3)},
    "lds_busy": round(c["SQ_LDS_IDX_ACTIVE"] / (cycles * 256.0), 3), "lds_conflict_frac": round(c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]), 3),
    "FETCH_SIZE_KiB": c["FETCH_SIZE"], "WRITE_SIZE_KiB": c["WRITE_SIZE"], "hbm_bytes": int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
    "l2_hit_rate": round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3),
    "source": f"rocprofv3 --pmc, one counter group per run (scripts/profile_pmc.sh TAG={tag}); summary kept as profiles/{R}_pmc_{tag}.txt",
}
out = os.path.join(root, "profiles", f"{R}_pmc.json")
data = json.load(open(out)) if os.path.exists(out) else {}
data[key] = rec
json.dump(data, open(out, "w"), indent=1)
open(os.path.join(root, "profiles", f"{R}_pmc_{tag}.txt"), "w").write(text)
print(key, json.dumps(rec, indent=1))
