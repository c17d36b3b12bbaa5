"""The drop-in boundary is a plain C ABI: include/taxor_gpu.h must compile as C11 (no C++), and a C program that
references every declared entry point must link against libtaxor_gpu.so."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_is_plain_c_and_links(tmp_path):
    hdr = os.path.join(ROOT, "include", "taxor_gpu.h")
    names = sorted(set(re.findall(r"\b(taxor_[a-z0-9_]+)\s*\(", open(hdr).read())) - {"taxor_status"})
    src = tmp_path / "abi.c"
    body = "\n".join(f"    p[{i}] = (fn)&{n};" for i, n in enumerate(names))
    src.write_text(f'#include "taxor_gpu.h"\n#include <stdio.h>\ntypedef void (*fn)(void);\nint main(void) {{\n'
                   f'    fn p[{len(names)}];\n{body}\n    printf("%d %d\\n", {len(names)}, p[0] != 0);\n    return 0;\n}}\n')
    exe = tmp_path / "abi"
    lib_dir = os.path.join(ROOT, "taxor_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", lib_dir, "-ltaxor_gpu", f"-Wl,-rpath,{lib_dir}",
                           "-Wl,--unresolved-symbols=ignore-in-shared-libs"])
    assert os.path.exists(exe)
