"""The library's measurement knobs sit behind ONE gate (taxor_amd/csrc/tuning.h): without TAXOR_TUNING=1 a stray TAXOR_QUERY_* /
TAXOR_SYNC_* / ... variable in a user's environment must not change the performance profile of a process that loads
libtaxor_gpu.so.  CPU side: (a) no source of the product reads a TAXOR_ variable except through the gate, (b) the gate itself
answers "not set" unless TAXOR_TUNING=1.  The behavioural check -- a searcher created under TAXOR_QUERY_PRUNE=0 still prunes --
needs kernels and lives in tests/test_gpu_parity.py."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "taxor_amd", "csrc")


def test_no_product_source_reads_a_knob_around_the_gate():
    offenders = []
    for fn in sorted(os.listdir(CSRC)):
        if not fn.endswith((".hip", ".cpp", ".h")) or fn == "tuning.h":
            continue
        for i, line in enumerate(open(os.path.join(CSRC, fn), errors="replace"), 1):
            code = line.split("//")[0]
            for m in re.finditer(r'(?<![A-Za-z_])getenv\s*\(\s*"([A-Z_0-9]+)"', code):
                if m.group(1).startswith("TAXOR_"):
                    offenders.append(f"{fn}:{i}: getenv(\"{m.group(1)}\")")
    assert not offenders, "\n".join(offenders)
    gated = sum(len(re.findall(r'tune_env\s*\(\s*"TAXOR_', open(os.path.join(CSRC, fn), errors="replace").read()))
                for fn in os.listdir(CSRC) if fn.endswith((".hip", ".cpp")))
    assert gated >= 40          # the knobs still exist, all of them behind the gate


def test_gate_answers_only_under_taxor_tuning(tmp_path):
    src = tmp_path / "gate.cpp"
    src.write_text('#include "tuning.h"\n#include <cstdio>\nint main() { const char *e = taxor::tune_env("TAXOR_QUERY_PRUNE"); '
                   'std::printf("%s\\n", e ? e : "unset"); return 0; }\n')
    exe = tmp_path / "gate"
    subprocess.check_call(["g++", "-std=c++17", "-I", CSRC, str(src), "-o", str(exe)])
    base = {k: v for k, v in os.environ.items() if not k.startswith("TAXOR_")}

    def run(**env):
        return subprocess.run([str(exe)], env={**base, **env}, capture_output=True, text=True, check=True).stdout.strip()

    assert run(TAXOR_QUERY_PRUNE="0") == "unset"
    assert run(TAXOR_QUERY_PRUNE="0", TAXOR_TUNING="0") == "unset"
    assert run(TAXOR_QUERY_PRUNE="0", TAXOR_TUNING="1") == "0"
    assert run(TAXOR_TUNING="1") == "unset"
