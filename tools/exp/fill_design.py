"""Fill the @PLACEHOLDERS@ of DESIGN.md section 6 from a collect_profiles.sh output directory:  python3 tools/exp/fill_design.py gpurun_out/r05c"""
import json, sys, collections, os
O = sys.argv[1]
def line(f):
    return json.loads(open(os.path.join(O, f)).read().strip().split("\n")[-1])
msg, ssg, sa, c5m, c5s = (line(f) for f in ("bench_msg.json", "bench_ssg.json", "bench_sa.json", "cfg5_msg.json", "cfg5_ssg.json"))
def pts(d):
    return "%.1f M" % (d["value"] / 1e6)
rep = {"@MSG@": "%.2f" % msg["ms_per_step"], "@MSGPTS@": pts(msg), "@SSG@": "%.2f" % ssg["ms_per_step"], "@SSGPTS@": pts(ssg),
       "@SA@": "%.2f" % sa["ms_per_step"], "@SAPTS@": pts(sa), "@C5S@": "%.2f" % c5s["ms_per_step"], "@C5SPTS@": pts(c5s),
       "@C5M@": "%.1f" % c5m["ms_per_step"], "@C5MPTS@": pts(c5m)}
r4 = {"pn2_conv1x1_fwd": 0.51, "pn2_conv1x1_bwd": 0.53, "pn2_conv1x1_dgrad": 0.43, "pn2_conv1x1_bwd_pair": 0.38, "pn2_conv1x1_wgrad": 0.38}
rows = []
for f in sorted(msg["roofline"]["families"], key=lambda x: -x["ms_per_step"]):
    tr = f.get("traffic")
    rows.append("| `%s` | %.2f | %.2f → **%.2f** | %.2f | %s |" % (f["name"], f["ms_per_step"], r4.get(f["name"], 0), f["frac"], f["hbm_frac"],
                ("%.2f×" % (tr / f["alg_bytes"])) if tr else "—"))
rep["@FAMROWS@"] = "\n".join(rows)
ab = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(os.path.join(O, "ab_switches.txt")):
    v, w, ms = l.split()
    ab[v][w].append(float(ms))
def arm(v):
    return "%s | %s" % (" / ".join("%.2f" % x for x in ab[v]["msg"]), " / ".join("%.2f" % x for x in ab[v]["ssg"]))
rep.update({"@AB_ON@": arm("PN2_SPLIT=1"), "@AB_SPLIT0@": arm("PN2_SPLIT=0"), "@AB_RES0@": arm("PN2_SPLIT_RES=0"), "@AB_NARROW0@": arm("PN2_SPLIT_NARROW=0"),
            "@AB_WGRAD0@": arm("PN2_SPLIT_WGRAD=0"), "@AB_K2560@": arm("PN2_SPLIT_K256=0"), "@AB_LOSS@": arm("PN2_BENCH_FORK=loss")})
s = open("DESIGN.md").read()
for k, v in rep.items():
    s = s.replace(k, v)
open("DESIGN.md", "w").write(s)
print({k: v for k, v in rep.items() if k != "@FAMROWS@"})
print(rep["@FAMROWS@"])
