"""Copy the round's measurement files from gpurun_out/r03p into profiles/ (attaching the PMC entries to the bench lines the
way bench.py does) and print the values DESIGN.md section 4 / README.md quote, for the documents to be updated from."""
import csv, glob, json, os, re, shutil, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(root, "gpurun_out", "r03p")
P = os.path.join(root, "profiles")


def load(name):
    lines = [l for l in open(os.path.join(O, name)) if l.startswith("{")]
    return json.loads(lines[-1])


for tag, key in (("book1_f64", "book1_1920x1080_spp512_f64"), ("teapot_f64", "teapot_1920x1080_spp1024_f64"),
                 ("million_f64", "million_3840x2160_spp256_f64"), ("movie_f64", "movie_1920x1080_spp512_f64"), ("book1_f32", "book1_1920x1080_spp512_f32")):
    subprocess.check_call([sys.executable, os.path.join(root, "scripts", "make_profile_json.py"), tag, key, "2"], stdout=subprocess.DEVNULL)
pmc = json.load(open(os.path.join(P, "r03_pmc.json")))
# bench.json was produced before r03_pmc.json existed on the box: attach the profile entries the way bench.py does
for src, dst in (("bench.json", "r03_bench.json"), ("bench_f32.json", "r03_bench_f32.json"), ("bench_teapot.json", "r03_bench_teapot.json"),
                 ("bench_million.json", "r03_bench_million.json"), ("bench_movie.json", "r03_bench_movie.json")):
    d = load(src)
    W, H = d["config"]["image"]
    wl = {"book1": "book1", "teapot.obj": "teapot", "1,000,001": "million", "teapot orbit": "movie"}
    name = next(v for k, v in wl.items() if d["config"]["workload"].startswith(k))
    key = f"{name}_{W}x{H}_spp{d['config']['spp']}_{d['dtype']}"
    if key in pmc:
        d["roofline"]["traffic"] = pmc[key]["hbm_bytes"]
        d["roofline"]["executed"] = dict(pmc[key], from_committed_profile=f"profiles/r03_pmc.json[{key}] -- collected in separate rocprofv3 --pmc "
                                                                            "runs of this workload, NOT measured in this run")
        k_ms = d["roofline"]["kernel_ms"]
        d["roofline"]["hbm"]["measured_GBps_from_profile"] = round(pmc[key]["hbm_bytes"] / (k_ms * 1e-3) / 1e9, 1)
    json.dump(d, open(os.path.join(P, dst), "w"))
    open(os.path.join(P, dst), "a").write("\n")
ks = max(glob.glob(os.path.join(O, "kt", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)   # the latest run's
shutil.copy(ks, os.path.join(P, "r03_bench_kernel_stats.csv"))
kt = {r["Name"]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(ks))}
kt_main = next(v for k, v in kt.items() if "pathtrace_kernel<double" in k)
kt_fin = next(v for k, v in kt.items() if "fx_finalize_kernel<double" in k)

b, b32, bt, bm, bv = (json.load(open(os.path.join(P, f))) for f in ("r03_bench.json", "r03_bench_f32.json", "r03_bench_teapot.json",
                                                                     "r03_bench_million.json", "r03_bench_movie.json"))


def pm(key):
    e = pmc[key]
    return (f"{e['valu_instr_per_simd_cycle']:.3f}, {e['lane_utilisation'] * 100:.0f} %, {e['wave_time_split']['waiting_on_counters'] * 100:.0f} %, "
            f"{e['l2_hit_rate'] * 100:.0f} % (lane-roofline {e['valu_lane_roofline_frac']:.2f})")


def tf(d):
    return f"{d['roofline']['achieved']:.2f} ({d['roofline']['frac']:.3f})"


vals = {
    "C2": f"{b['value']:.0f}", "C2MS": f"{b['ms_per_step']:.1f}", "C2F32": f"{b32['value']:.0f}", "C2F32MS": f"{b32['ms_per_step']:.1f}",
    "C2F32TF": f"{b32['roofline']['achieved']:.2f} ({b32['roofline']['frac']:.3f}", "TFLOPS": f"{b['roofline']['achieved']:.2f}",
    "FRAC": f"{b['roofline']['frac']:.3f}", "FRAC_NOFMA": f"{b['roofline']['frac_of_peak_without_fma']:.3f}",
    "C3": f"{bt['value']:.0f}", "C3MS": f"{bt['ms_per_step']:.0f}", "C3TF": tf(bt), "C4": f"{bm['value']:.0f}", "C4MS": f"{bm['ms_per_step']:.0f}",
    "C4TF": tf(bm), "C5": f"{bv['value']:.0f}", "C5MS": f"{bv['ms_per_step']:.0f}", "C5TF": tf(bv),
    "CPU_SANE": f"{b['cpu_baseline']['value']:.1f}", "CPU_FAITH": f"{b['cpu_baseline_faithful']['value']:.2f}",
    "HBM_GB": f"{pmc['book1_1920x1080_spp512_f64']['hbm_bytes'] / 1e9:.0f}",
    "HBM_TBS": f"{pmc['book1_1920x1080_spp512_f64']['hbm_bytes'] / (b['roofline']['kernel_ms'] * 1e-3) / 1e12:.2f}",
    "PMC_C2": pm("book1_1920x1080_spp512_f64"), "PMC_C3": pm("teapot_1920x1080_spp1024_f64"), "PMC_C4": pm("million_3840x2160_spp256_f64"),
    "PMC_C5": pm("movie_1920x1080_spp512_f64"), "PMC_C2F32": pm("book1_1920x1080_spp512_f32"), "KT_MS": f"{kt_main:.1f}", "KT_FIN": f"{kt_fin:.2f}",
}
print(json.dumps(vals, indent=1))
