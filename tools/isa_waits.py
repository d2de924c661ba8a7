"""Load / wait / branch skeleton of one kernel from a device-only assembly dump:
    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -w -S --cuda-device-only -o /tmp/mmf.s multimotionfusion_amd/csrc/mmf_hip.hip
    python tools/isa_waits.py /tmp/mmf.s clean_flag_kernel
A `s_waitcnt vmcnt(0)` between two groups of loads is a dependent memory round trip."""
import re
import sys

s = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    m = re.search(r"; -- Begin function (_ZN3mmf\d+%s\S*)" % name, s)
    i = m.start()
    j = s.index(".amdhsa_kernel", i)
    n = 0
    print("==", name)
    for ln in s[i:j].splitlines():
        t = ln.strip()
        if not t or t.startswith(";"):
            continue
        n += 1
        if re.match(r"(global_load|global_store|global_atomic|buffer_|s_waitcnt|s_barrier|ds_|s_cbranch|\.LBB|s_endpgm|scratch_|s_load)", t):
            print(n, t.split(";")[0])
