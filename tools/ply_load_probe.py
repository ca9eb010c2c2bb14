"""How long does the .ply path take at real size, and how much memory?  (ResourceManager::loadGaussians,
Engine/ResourceManager.cpp:167-300 -> gs_convert_ply / gs_load_ply.)

Writes a synthetic cloud of a BASELINE config (default C: 5,834,784 gaussians, the Garden-30k shape) as a binary
little-endian .ply with the INRIA property set (62 float properties per vertex incl. normals: 1.45 GB), then converts it
in a fresh child process -- no GPU needed, gs_convert_ply is host code -- and reports seconds and peak RSS.

    python tools/ply_load_probe.py [C] [--keep]
"""
import argparse, ctypes as C, json, os, resource, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def write_ply(path, aos):
    """The inverse of ResourceManager.cpp:229-273 (to float32 rounding): records -> INRIA .ply rows."""
    n = aos.shape[0]
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
            ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    rows = np.zeros((n, len(names)), "<f4")
    col = {k: i for i, k in enumerate(names)}
    rows[:, col["x"]], rows[:, col["y"]], rows[:, col["z"]] = -aos[:, 0], -aos[:, 1], aos[:, 2]
    for a in range(3):
        rows[:, col[f"scale_{a}"]] = np.log(aos[:, 4 + a])
        rows[:, col[f"f_dc_{a}"]] = aos[:, 12 + a]
    rows[:, col["rot_0"]], rows[:, col["rot_1"]] = aos[:, 10], -aos[:, 11]
    rows[:, col["rot_2"]], rows[:, col["rot_3"]] = -aos[:, 8], -aos[:, 9]
    a = np.clip(aos[:, 15].astype(np.float64), 1e-7, 1 - 1e-7)
    rows[:, col["opacity"]] = np.log(a / (1 - a))
    for c in range(15):
        for ch in range(3):
            rows[:, col[f"f_rest_{c + 15 * ch}"]] = aos[:, 16 + 4 * c + ch]
    with open(path, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\n" + f"element vertex {n}\n".encode())
        for k in names:
            f.write(f"property float {k}\n".encode())
        f.write(b"end_header\n")
        rows.tofile(f)
    return os.path.getsize(path)


def _hwm():
    """peak resident set of THIS process image (ru_maxrss survives fork + exec and would report the parent's peak)"""
    with open("/proc/self/status") as f:
        for line in f:
            if line.startswith("VmHWM:"):
                return int(line.split()[1]) * 1024
    return 0


def child(path):
    from vk3dgaussiansplatting_amd import _lib
    L = _lib.lib()
    with open("/proc/self/clear_refs", "w") as f:
        f.write("5")                                                           # reset the high-water mark
    base_rss = _hwm()                                                          # interpreter + numpy + the library
    n = C.c_uint32()
    t0 = time.perf_counter()
    assert L.gs_convert_ply(os.fsencode(path), None, 0, C.byref(n)) == 0          # header only
    t_hdr = time.perf_counter() - t0
    out = np.empty((n.value, 84), np.float32)
    t0 = time.perf_counter()
    rc = L.gs_convert_ply(os.fsencode(path), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n))
    secs = time.perf_counter() - t0
    assert rc == 0, L.gs_ply_last_error()
    rss = _hwm()                                                               # before the sanity checks below allocate anything
    print(json.dumps({"gaussians": int(n.value), "header_only_s": round(t_hdr, 4), "convert_s": round(secs, 2),
                      "peak_rss_bytes": rss, "baseline_rss_bytes": base_rss, "output_bytes": int(out.nbytes),
                      "opacity_mean": float(out[:, 15].mean()), "finite": bool(np.isfinite(out[::97]).all())}))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    ap = argparse.ArgumentParser()
    ap.add_argument("config", nargs="?", default="C")
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    from vk3dgaussiansplatting_amd import synth
    aos = synth.generate_config(a.config)[0]
    d = tempfile.mkdtemp(prefix="gs_ply_", dir="/tmp")
    path = os.path.join(d, f"config_{a.config}.ply")
    size = write_ply(path, aos)
    del aos
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path], capture_output=True, text=True, check=True).stdout
    res = json.loads(out.strip().splitlines()[-1])
    res.update(file_bytes=size, rss_over_file=round(res["peak_rss_bytes"] / size, 2),
               conversion_rss_over_file=round((res["peak_rss_bytes"] - res["baseline_rss_bytes"]) / size, 2), host_cpus=os.cpu_count())
    print(json.dumps(res))
    if not a.keep:
        os.remove(path)
        os.rmdir(d)
