"""Runs the kernel source on the CPU through tests/emu (one host thread per lane, ASan on)."""
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EMU_DIR = os.path.join(HERE, "emu")
EXE = os.path.join(EMU_DIR, "emu_run")


def build():
    subprocess.check_call(["make", "-s", "-C", EMU_DIR])


def run(src, src_off, dst_off, flags=0, lanes=64, variant=0, timeout=600, plain=False, check=0):
    build()
    n = len(src_off) - 1
    src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
    dst_off = np.ascontiguousarray(dst_off, dtype=np.uint64)
    src = np.ascontiguousarray(src, dtype=np.uint8)[: int(src_off[-1])]
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.uint64(n).tobytes() + src_off.tobytes() + dst_off.tobytes() + src.tobytes())
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
        p = subprocess.run([EXE, fin, fout, str(lanes), str(flags), str(variant), "1" if plain else "0", str(check)],
                           env=env, timeout=timeout,
                           capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("emu_run failed (%d): %s" % (p.returncode, p.stderr[-3000:]))
        raw = open(fout, "rb").read()
    status = np.frombuffer(raw, dtype=np.int32, count=n, offset=0)
    outlen = np.frombuffer(raw, dtype=np.uint32, count=n, offset=4 * n)
    detail = np.frombuffer(raw, dtype=np.int32, count=n, offset=8 * n)
    inpos = np.frombuffer(raw, dtype=np.uint32, count=n, offset=12 * n)
    dst = np.frombuffer(raw, dtype=np.uint8, offset=16 * n)
    return dst, status, outlen, detail, inpos


def run_resume(src, src_off, dst_off, flags=0, chunk=1, timeout=900):
    """Every stream fed to the incremental (infl_stream) kernel in pieces of `chunk` bytes.
    -> (dst, status, outlen): status 0 = final block complete, else the last call's detail code."""
    build()
    n = len(src_off) - 1
    src_off = np.ascontiguousarray(src_off, dtype=np.uint64)
    dst_off = np.ascontiguousarray(dst_off, dtype=np.uint64)
    src = np.ascontiguousarray(src, dtype=np.uint8)[: int(src_off[-1])]
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.uint64(n).tobytes() + src_off.tobytes() + dst_off.tobytes() + src.tobytes())
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
        p = subprocess.run([EXE, fin, fout, "resume", str(flags), str(chunk), "0"], env=env, timeout=timeout,
                           capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("emu_run resume failed (%d): %s" % (p.returncode, p.stderr[-3000:]))
        raw = open(fout, "rb").read()
    status = np.frombuffer(raw, dtype=np.int32, count=n, offset=0)
    outlen = np.frombuffer(raw, dtype=np.uint32, count=n, offset=4 * n)
    dst = np.frombuffer(raw, dtype=np.uint8, offset=8 * n)
    return dst, status, outlen


def run_png(filtered, geom, in_len=None, in_status=None, timeout=900):
    """The PNG un-filter kernel on a list of filtered images; geom = [(rows, rowbytes, bpp)].
    -> (list of rebuilt images, status)"""
    build()
    n = len(filtered)
    io = np.zeros(n + 1, dtype=np.uint64)
    oo = np.zeros(n + 1, dtype=np.uint64)
    io[1:] = np.cumsum([len(f) for f in filtered], dtype=np.uint64)
    oo[1:] = np.cumsum([r * rb for r, rb, _ in geom], dtype=np.uint64)
    g = np.zeros((n, 4), dtype=np.uint32)
    for i, (rows, rowbytes, bpp) in enumerate(geom):
        g[i, :3] = rows, rowbytes, bpp
    in_len = np.array([len(f) for f in filtered] if in_len is None else in_len, dtype=np.uint32)
    in_status = np.zeros(n, dtype=np.int32) if in_status is None else np.array(in_status, dtype=np.int32)
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.uint64(n).tobytes() + io.tobytes() + oo.tobytes() + g.tobytes() + in_len.tobytes() +
                    in_status.tobytes() + b"".join(filtered))
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
        p = subprocess.run([EXE, fin, fout, "png", "0", "0", "0"], env=env, timeout=timeout, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("emu_run png failed (%d): %s" % (p.returncode, p.stderr[-3000:]))
        raw = open(fout, "rb").read()
    status = np.frombuffer(raw, dtype=np.int32, count=n, offset=0)
    out = raw[4 * n:]
    return [out[int(oo[i]):int(oo[i + 1])] for i in range(n)], status


def run_gather(blob, piece_off, piece_len, piece_dst, out_len, timeout=600, want_status=False):
    """infl_gather_kernel in the emulator, told the sizes of blob and arena: -> the gathered arena (uint8, 0xee where
    nothing was written), with want_status also the per-piece status (0, or -1 for a piece outside blob or arena)."""
    n = len(piece_off)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([n, len(blob), out_len], dtype=np.uint64).tobytes())
            f.write(np.asarray(piece_off, dtype=np.uint64).tobytes())
            f.write(np.asarray(piece_dst, dtype=np.uint64).tobytes())
            f.write(np.asarray(piece_len, dtype=np.uint32).tobytes())
            f.write(bytes(blob))
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0")
        p = subprocess.run([EXE, fin, fout, "gather", "0", "0"], capture_output=True, timeout=timeout, env=env)
        if p.returncode != 0:
            raise RuntimeError("emu_run gather failed (%d): %s" % (p.returncode, p.stderr.decode()[-2000:]))
        raw = np.fromfile(fout, dtype=np.uint8)
        out, st = raw[:out_len], raw[out_len:out_len + 4 * n].view(np.int32)
        return (out, st) if want_status else out
