"""Is k_count near the floor of ANY kernel that reads its words once?  Read-only stream kernels (k_stream_read<uint4>: 16-byte loads,
a checksum per thread, nothing stored) over Count's footprint at config C -- 13,121,624 sixteen-bit words (k_count<true>) and as many
thirty-two-bit words (k_count<false>) -- at several grids, the buffer warm in the Infinity Cache (the state Count finds its words in,
minus the part the preceding Scatter left in L2), and, in the same process, config C's sort so that k_count appears in the same trace.
Run under rocprofv3 --kernel-trace --stats (tools/count_floor.sh); this script prints the event-timed figures (launch gaps included)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vk3dgaussiansplatting_amd as gs
from vk3dgaussiansplatting_amd import _lib
L = _lib.lib()
h = C.c_void_p(); assert L.gs_create(None, C.byref(h)) == 0
E = 13_121_624
print("# read-only kernel, 256-thread blocks, 200 launches back to back (hipEvents around the loop: launch gaps included)")
print("#    bytes   blocks   us/launch   GB/s")
for nbytes in (2 * E, 4 * E):
    for blocks in (512, 1024, 2048, 4096, 8192):
        g, ms = C.c_float(), C.c_float()
        rc = L.gs_membench(h, 0, nbytes, blocks, 200, C.byref(g), C.byref(ms))
        assert rc == 0
        print(f"{nbytes:10d} {blocks:8d} {ms.value * 1e3:10.2f} {g.value:8.0f}", flush=True)
L.gs_destroy(h)
rs = gs.RadixSort(count_launches=gs.GS_COUNT_PER_PASS)
rs.initForScene(E, 8160)
ms, ok = rs.bench(E, 8160, iters=20)
print(f"# stand-alone sort of {E} random keys, 48 bits (12 passes, 32-bit words throughout): {ms:.4f} ms, sorted={ok}")
rs.cleanup()
