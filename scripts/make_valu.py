"""profiles/valu.json from the SQ counter passes of scripts/profile_pmc.sh (summarised by summarize_pmc.py).

  valu_wave_instr_per_simd_cycle = SQ_INSTS_VALU / (shader cycles * 1024 SIMDs); shader cycles = GRBM_GUI_ACTIVE / 8 XCDs
  busy fraction                  = that / the issue peak measured with scripts/calib/valu_peak.hip at 4 waves per SIMD
  lane_utilisation               = SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64)
  wave_time_split                = issuing: SQ_ACTIVE_INST_ANY; issue_stalled: SQ_WAIT_INST_ANY - SQ_WAIT_INST_LDS;
                                   waiting_on_counters: the remainder -- all per SQ_WAVE_CYCLES
  lds_busy                       = SQ_LDS_IDX_ACTIVE / (shader cycles * 256 CUs); lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
usage: make_valu.py SUMMARY.txt KEY KERNEL_MS SAMPLES
"""
import json, os, re, sys

text, key, kernel_ms, samples = open(sys.argv[1]).read(), sys.argv[2], float(sys.argv[3]), float(sys.argv[4])


def get(name):
    v = [float(x) for x in re.findall(r'"%s": ([0-9.e+]+)' % name, text)]
    return sum(v) / len(v)


PEAK_4_WAVES = 0.3215   # wave64 FMA issues per SIMD cycle, scripts/calib/valu_peak.hip (3.1 cycles per independent FMA)
cycles = get("GRBM_GUI_ACTIVE") / 8.0
valu = get("SQ_INSTS_VALU") / (cycles * 1024.0)
wave_cycles = get("SQ_WAVE_CYCLES")
issuing = get("SQ_ACTIVE_INST_ANY") / wave_cycles
waiting = (get("SQ_WAIT_INST_ANY") - get("SQ_WAIT_INST_LDS")) / wave_cycles
out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "valu.json")
data = json.load(open(out_path)) if os.path.exists(out_path) else {}
data[key] = {
    "valu_wave_instr_per_simd_cycle": round(valu, 4),
    "measured_issue_peak_4_waves_per_simd": PEAK_4_WAVES,
    "valu_busy_frac_of_measured_peak": round(valu / PEAK_4_WAVES, 3),
    "lane_utilisation": round(get("SQ_THREAD_CYCLES_VALU") / (get("SQ_ACTIVE_INST_VALU") * 64.0), 3),
    "wave_time_split": {"issuing": round(issuing, 3), "waiting_on_counters": round(max(0.0, 1.0 - issuing - waiting), 3),
                        "issue_stalled": round(waiting, 3)},
    "lds_busy": round(get("SQ_LDS_IDX_ACTIVE") / (cycles * 256.0), 3),
    "lds_conflict_frac": round(get("SQ_LDS_BANK_CONFLICT") / get("SQ_LDS_IDX_ACTIVE"), 3),
    "shader_clock_ghz": round(cycles / (kernel_ms * 1e6), 2),
    "valu_lane_slots_per_sample": round(get("SQ_INSTS_VALU") * 64.0 / samples, 1),
    "source": "rocprofv3 --pmc passes in profiles/r01_final_pmc_512spp_f32.txt (scripts/make_valu.py); peak from "
              "scripts/calib/valu_peak.hip on the same chip (1/2/4/8 waves per SIMD: 7.5/4.2/3.1/2.8 cycles per independent wave64 FMA)",
}
json.dump(data, open(out_path, "w"), indent=1)
print(key, json.dumps(data[key], indent=1))
