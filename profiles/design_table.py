#!/usr/bin/env python3
"""DESIGN.md section 5's table from ONE bench.py line (the headline and the `layouts` legs of the same invocation).
usage: python profiles/design_table.py profiles/r06/bench_gtdb.json [profiles/r06/bench_build.json] [--write]
(--write: replace the block between the <!-- BENCH-TABLE --> markers of DESIGN.md; the second file adds the builder's line)"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = sys.argv[1]
j = json.loads([l for l in open(path) if l.startswith("{")][-1])
c, r = j["config"], j["roofline"]
lv = r.get("levels", [])
gc = r.get("gather_ceiling", {})
out = []
out.append(f"Source: `{os.path.relpath(path, ROOT)}` — one `python bench.py` invocation ({j['steps']} steps, {j['warmup']} warm-up; the driver's own run is `BENCH_r06.json`).")
out.append("")
out.append("| leg (same invocation) | index | reads | Mbp/s | `frac` | other fractions | work per read |")
out.append("|---|---|---|---|---|---|---|")
un = r.get("unpruned") or {}
out.append(f"| **headline `value`** (resident, {j['steps']} steps) | {c['index_bytes'] / 1e9:.0f} GB, root {c['root_bins']}, children {c['child_bins']}, depth {c['depth']} | "
           f"{c['reads_per_gpu']} × {c['read_len'] // 1000} kb, error {c['read_error']}, forward strand | **{j['value']:,.0f}** ({j['ms_per_step']:.1f} ms/step) | **{r['frac']:.3f}** | "
           f"contract {r.get('contract_frac')} (pruning off: {un.get('value_Mbp_s', 0):,.0f} Mbp/s), moved {r.get('moved_frac')}, line128 {r.get('requested_accounting', {}).get('frac_line128')} | "
           f"{c['hashes_per_read']} hashes, {c['work_items_per_read']} items, {c['tuples_per_read']} tuples |")
if "value_host_fed" in j:
    p = j.get("pcie_inclusive") or {}
    out.append(f"| host-fed (PCIe inside; never `value`) | same | ≥ {j.get('sustained', {}).get('reads', 0):,} reads through `taxor_gpu_search_batch` | {j['value_host_fed']:,.0f} sustained; {p.get('value', 0):,.0f} single pageable call | | | |")
if "read_error_0.04" in j:
    e = j["read_error_0.04"]
    out.append(f"| read error 0.04 (BASELINE.md §3; the easier case) | same | same shape | {e['value']:,.0f} ({e.get('value_host_fed', 0):,.0f} host-fed) | {e.get('frac')} | | {e['work_items_per_read']} items, {e['tuples_per_read']} tuples |")
for leg in j.get("layouts", []):
    if "error" in leg:
        out.append(f"| `layouts`: {leg['layout']} | (failed: {leg['error'][:80]}) | | | | | |")
        continue
    out.append(f"| `layouts`: {leg['layout']} — {leg['what'].split(':')[0].split(';')[0]} | {leg['index_bytes'] / 1e9:.0f} GB, root {leg['root_bins']}, children {leg['child_bins']}, depth {leg['depth']} | "
               f"frac_reverse {leg['frac_reverse']} | {leg['value']:,.0f} | {leg['frac']} | algorithmic {leg['algorithmic_frac']}; no PMC pass | "
               f"{leg['hashes_per_read']} hashes, {leg['work_items_per_read']} items, {leg['tuples_per_read']} tuples |")
cb = j.get("cpu_baseline") or {}
if cb:
    out.append(f"| CPU baseline ({cb.get('kind')}, {cb.get('cores')} cores, reference's `do_parallel` scheduler) | same | {cb.get('sample', '')[:60]}… | {cb.get('value', 0):,.0f} | | GPU results bit-identical on the sample | |")
out.append("")
if lv:
    out.append("Per level of the headline: " + "; ".join(
        f"level {x['level']} {x['ms_per_step']} ms/step, {x['requested_GBps']:,.0f} GB/s requested ({x['frac']}), {x['row_reads_G_per_s']} G rows/s of {x['bytes_per_row_read']:.0f} B" for x in lv) + ".")
if gc:
    out.append(f"Gather ceilings of the same run (random whole rows, nothing else): root {gc['root']['GBps']:,.0f} GB/s"
               + (f", children {gc['children']['GBps']:,.0f} GB/s" if "children" in gc else "") + ".")
if r.get("traffic") is not None:
    out.append(f"Memory side (live `rocprofv3 --pmc` passes in the same invocation): {r['traffic'] / 1e9:.2f} GB per launch = {r.get('traffic_GBps', 0):,.0f} GB/s "
               f"({r.get('traffic_over_requested')} × requested, {r.get('traffic_over_line128')} × the 128-B lines touched); average launch {r['avg_launch_ms']} ms "
               f"(rocprofv3 `--kernel-trace --stats` of the same command without the legs: `profiles/r06/gtdb_kernel_stats.csv`, `k_query_level<true,2,...>` average in its first row).")
extra = [a for a in sys.argv[2:] if not a.startswith("--")]
if extra:
    b = json.loads([l for l in open(extra[0]) if l.startswith("{")][-1])
    bc, br, bs = b["config"], b["roofline"], b["stage_s_per_step"]
    cbb = b.get("cpu_baseline") or {}
    out.append("")
    out.append(f"Index construction (`{os.path.relpath(extra[0], ROOT)}`, `python bench.py --mode build`, {b['steps']} steps = whole builds, {b['warmup']} warm-up): "
               f"**{b['value'] / 1e9:.2f} G key insertions/s** -- {bc['insertions_per_step'] / 1e9:.2f} G insertions per build of {bc['index_bytes'] / 1e9:.1f} GB "
               f"({bc['children']} children x {bc['child_bins']} bins x {bc['keys_per_bin']} keys under a root of merged bins, every bin built) in {bs['total']:.2f} s: "
               f"peel {bs['peel']:.2f} s, assign + verify {bs['assign_verify']:.2f} s, unions {bs['unions']:.2f} s; {bc['chunks_per_step']} chunks of <= 3 GB scratch, "
               f"{bc['rounds_max']} rounds at most, {bc['reseeds']} reseeds.  `roofline.frac` {br['frac']} of HBM bytes ({br['algorithmic_bytes_per_insertion']} B per insertion) -- not its bound; "
               f"`rmw`: {br['rmw']['achieved_G_per_s']} G random read-modify-writes/s = {br['rmw']['frac']} of the measured 27 G/s averaged over peel + assign + verify"
               + (f"; reference `AddAll` on one core: {cbb['value'] / 1e6:.1f} M insertions/s ({b['value'] / cbb['value']:.0f}x)" if cbb else "")
               + (f"; from keys in host memory (upload at {b['pcie_inclusive']['upload_GBps']} GB/s inside, never `value`): {b['value_host_fed'] / 1e9:.2f} G insertions/s" if 'pcie_inclusive' in b else "")
               + ".  Per kernel (HIP events inside the library): " + "; ".join(f"`{k['kernel']}` {k['achieved_G_per_s']} G RMW/s = {k['frac']} of the ceiling" for k in br['rmw'].get('kernels', []))
               + ".  Longest builder launch in the `rocprofv3 --kernel-trace --stats` of the same command: 3.2 ms (`k_assign`; `k_count_lds` 2.5, `k_round` 2.5, `k_count` 2.0, `k_set_mark` 1.3; "
                 "`profiles/r06/build_kernel_stats.csv`); one counter pass: `profiles/r06/build_pmc.txt`.")
text = "\n".join(out)
print(text)
if "--write" in sys.argv:
    dp = os.path.join(ROOT, "DESIGN.md")
    s = open(dp).read()
    block = "<!-- BENCH-TABLE -->\n" + text + "\n<!-- /BENCH-TABLE -->"
    if "<!-- /BENCH-TABLE -->" in s:
        s = re.sub(r"<!-- BENCH-TABLE -->.*?<!-- /BENCH-TABLE -->", lambda m: block, s, flags=re.S)
    else:
        s = s.replace("<!-- BENCH-TABLE -->", block)
    open(dp, "w").write(s)
