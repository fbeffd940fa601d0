"""CPU: the embedded text of the persistent kernel compiles under this machine's hiprtc (no GPU needed: hiprtc cross-compiles gfx950),
the code object lands in FDIFF_CACHE_DIR and is served from there the second time, and an instantiation the template rejects is
reported as FD_ERR_UNSUPPORTED with the compiler's message (the engine then keeps its run-time-shape kernel)."""
import ctypes as C
import glob


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
