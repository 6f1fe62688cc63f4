#!/usr/bin/env python3
"""Register / LDS / spill table of every HIP kernel of the engine (hipcc --offload-arch=gfx950 -S, no GPU needed):
    python profiles/scripts/kernel_resources.py [--f16] > profiles/rNN_kernel_resources.txt
Columns from the code-object metadata: vgpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count, private (scratch) bytes,
static LDS bytes. tests/test_kernel_resources.py asserts the scratch / VGPR-spill columns are zero."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "whisper.axera_amd", "csrc")
FILES = ["frontend", "gemm", "encoder_attn", "decoder", "decode_gemv", "decode_gemm", "decode_persistent", "decode_persistent2"]
f16 = "1" if "--f16" in sys.argv else "0"
print(f"# hipcc --offload-arch=gfx950 -O3 -DAXW_F16={f16}; kernel | vgpr | sgpr | vgpr spills | sgpr spills | scratch B | static LDS B")
for name in FILES:
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, name + ".s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), f"-DAXW_F16={f16}",
                        "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, name + ".hip")], check=True, capture_output=True)
        text = open(out).read()
    print(f"## {name}.hip")
    for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size:", text, re.S):
        def g(k):
            m = re.search(r"\." + k + r":\s+(\S+)", blk)
            return m.group(1) if m else "?"
        sym = g("name")
        try:
            dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip() or sym
        except OSError:
            dem = sym
        dem = re.sub(r"\(.*", "", dem).replace("void ", "").replace("axw::bf::", "").replace("axw::hf::", "")
        print(f"{dem} | {g('vgpr_count')} | {g('sgpr_count')} | {g('vgpr_spill_count')} | {g('sgpr_spill_count')} | {g('private_segment_fixed_size')} | {g('group_segment_fixed_size')}")
