"""HIP API time on the host between the last two observed-class sweeps (= the latent-class part of the last full
iteration), from rocprofv3 --hip-trace --kernel-trace --output-format csv.
usage: python profiles/hip_api_window.py <dir with *_hip_api_trace.csv and *_kernel_trace.csv>"""
import glob
import sys

import pandas as pd

d = sys.argv[1]
k = pd.read_csv(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0])
a = pd.read_csv(glob.glob(d + "/**/*_hip_api_trace.csv", recursive=True)[0])
fc = k[k["Kernel_Name"].str.contains("final_choice_kernel")].sort_values("Start_Timestamp")
t0, t1 = fc["End_Timestamp"].iloc[-2], fc["Start_Timestamp"].iloc[-1]
w = a[(a["Start_Timestamp"] >= t0) & (a["End_Timestamp"] <= t1)].copy()
w["dur"] = w["End_Timestamp"] - w["Start_Timestamp"]
print(f"window {1e-6 * (t1 - t0):.1f} ms, {len(w)} HIP API calls, {1e-6 * w['dur'].sum():.1f} ms inside them")
g = w.groupby("Function")["dur"].agg(["count", "sum", "max"]).sort_values("sum", ascending=False)
for name, r in g.head(14).iterrows():
    print(f"{1e-6 * r['sum']:9.2f} ms  x{int(r['count']):6d}  max {1e-3 * r['max']:9.1f} us  {name}")
