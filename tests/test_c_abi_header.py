"""The drop-in boundary is a plain C ABI: include/taxor_gpu.h (the seam that replaces hixf::do_parallel and what a binding needs
around it) and include/taxor_gpu_tools.h (everything else the library exports) must each compile as C11 (no C++), and a C program
that references every entry point a header declares must link against libtaxor_gpu.so.  The seam stays small enough to read."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = re.sub(r"/\*.*?\*/", " ", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(taxor_[a-z0-9_]+)\s*\(", text)) - {"taxor_status"})


@pytest.mark.parametrize("header", ["taxor_gpu.h", "taxor_gpu_tools.h"])
def test_header_is_plain_c_and_links(tmp_path, header):
    names = declared(header)
    assert len(names) >= 20
    src = tmp_path / "abi.c"
    body = "\n".join(f"    p[{i}] = (fn)&{n};" for i, n in enumerate(names))
    src.write_text(f'#include "{header}"\n#include <stdio.h>\ntypedef void (*fn)(void);\nint main(void) {{\n'
                   f'    fn p[{len(names)}];\n{body}\n    printf("%d %d\\n", {len(names)}, p[0] != 0);\n    return 0;\n}}\n')
    exe = tmp_path / "abi"
    lib_dir = os.path.join(ROOT, "taxor_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", lib_dir, "-ltaxor_gpu", f"-Wl,-rpath,{lib_dir}",
                           "-Wl,--unresolved-symbols=ignore-in-shared-libs"])
    assert os.path.exists(exe)


def test_the_seam_is_small_and_free_of_tools():
    """taxor_gpu.h is what a maintainer of the reference reads: at most 250 lines, no measurement aids, builders, variant scans,
    synthetic reads or inflate entry points (those are taxor_gpu_tools.h), and the two headers declare disjoint sets"""
    seam, tools = declared("taxor_gpu.h"), declared("taxor_gpu_tools.h")
    assert len(open(os.path.join(ROOT, "include", "taxor_gpu.h")).read().splitlines()) <= 250
    assert not set(seam) & set(tools)
    for n in seam:
        assert not re.search(r"gather_ceiling|gather_pattern|phase_profile|variant|synth|fill_random|inflate|build_|probe", n), n
    for n in ("taxor_gpu_index_create", "taxor_gpu_index_create_replicated", "taxor_gpu_searcher_create", "taxor_gpu_search_batch",
              "taxor_gpu_search_batch_begin", "taxor_gpu_search_batch_end", "taxor_gpu_search_segments_begin", "taxor_gpu_gather_results",
              "taxor_gpu_last_error", "taxor_threshold_select", "taxor_hixf_load", "taxor_format_reads"):
        assert n in seam, n
