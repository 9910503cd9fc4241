#!/usr/bin/env python3
"""Static smell test for kernels that wait instead of computing (no GPU needed).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -fno-slp-vectorize -S --cuda-device-only \
          -o /tmp/isa_X.s dctn_amd/csrc/X.hip
    python tools/scan_waited_loads.py /tmp/isa_*.s

Per kernel: vector-memory loads that are followed within three instructions by `s_waitcnt vmcnt(0)` ("waited on the spot":
a memory round trip with nothing behind it), all vector-memory loads, scratch accesses (a spill shares the loads' in-order
counter and turns every wait into vmcnt(0)), instructions.  Round 5 found the fold backward's per-step waits, the linear
head's branch per weight row and the flag scans this way (NOTEBOOK)."""
import re
import sys

LOAD = re.compile(r"(global_load|buffer_load|flat_load)")


def kernels(path):
    name, lines = None, []
    for line in open(path):
        m = re.match(r"^(_Z[A-Za-z0-9_]+):", line)
        if m:
            name, lines = m.group(1), []
        elif name:
            lines.append(line)
            if "s_endpgm" in line:
                yield name, lines
                name, lines = None, []


def main():
    rows = []
    for path in sys.argv[1:]:
        for name, lines in kernels(path):
            ins = [l.strip() for l in lines if l.startswith("\t") and not l.strip().startswith((".", ";"))]
            waited = 0
            for k, i in enumerate(ins):
                if LOAD.match(i):
                    for j in range(k + 1, min(k + 4, len(ins))):
                        if LOAD.match(ins[j]):
                            break
                        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", ins[j])
                        if m and int(m.group(1)) == 0:
                            waited += 1
                            break
            rows.append((waited, sum(1 for i in ins if LOAD.match(i)), sum(1 for i in ins if i.startswith("scratch_")), len(ins),
                         name))
    rows.sort(reverse=True)
    print("waited  loads scratch   insts  kernel")
    for w, n, sc, total, name in rows[:40]:
        print(f"{w:6d} {n:6d} {sc:7d} {total:7d}  {name[:140]}")


if __name__ == "__main__":
    main()
