#!/bin/bash
# builds tools/gather_roof (gfx950; cross-compiles without a GPU): the memory-side roof of the config-5 RK45 access pattern (gather_roof.hip)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 gather_roof.hip -o gather_roof && echo built tools/gather_roof
