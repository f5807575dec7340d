#!/bin/sh
# builds tools/kbench against the in-tree library
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/kbench.cpp -Luzu_amd/lib -luzu_hip -Wl,-rpath,'$ORIGIN/../uzu_amd/lib' -o tools/kbench
