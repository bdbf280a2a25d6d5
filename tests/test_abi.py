"""The C-ABI library loads and exports every symbol include/tinyrenderer_hip.h declares; with
no GPU present the product fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

from tinyrenderer_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "tinyrenderer_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    lib = _ffi.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(_ffi.EXPORTS) == names
    assert b"gfx950" in lib.tr_version()


def test_no_silent_fallback_without_gpu():
    import torch

    if torch.cuda.is_available() or os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    lib = _ffi.load()
    ctx = C.c_void_p()
    assert lib.tr_create(0, 64, 64, C.byref(ctx)) == -3  # TR_ERR_NO_DEVICE
    assert not ctx.value
    from tinyrenderer_amd.model_renderer import ModelRendererState

    with pytest.raises(_ffi.TrError):
        ModelRendererState(64, 64)


def test_argument_validation_without_gpu():
    lib = _ffi.load()
    assert lib.tr_create(0, 0, 10, C.byref(C.c_void_p())) == -6  # TR_ERR_SIZE
    assert lib.tr_clear(None) == -1
    assert lib.tr_draw(None, None, None) == -1
    assert lib.tr_last_error(None) == b"null context"


def test_product_does_not_reference_the_oracle():
    """The product path must never import, link, load or execute anything under oracle/."""
    pkg = os.path.join(ROOT, "tinyrenderer_amd")
    banned = ("liboracle", "oracle_ffi", "import oracle", "from oracle", "oracle/", "tr_oracle")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="replace").read().lower()
                for b in banned:
                    assert b not in text, f"{os.path.join(dirpath, f)} references {b!r}"


def test_release_library_reads_only_the_documented_switches():
    """The release build's environment surface is the table of INTEGRATION.md 4.1 and nothing else: the experiment,
    trace and wrong-pixel switches live in the diagnostic build (`make diag`, -DTR_DIAG) only."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    start = text.index("### 4.1 Environment switches")
    table = text[start:text.index("**Diagnostic build.**", start)]
    documented = set(re.findall(r"`(TR_[A-Z0-9_]+)`", table))
    assert {"TR_SHADE_EXACT", "TR_RASTER_MODE", "TR_TILE"} <= documented and len(documented) <= 12

    def switches(path):
        blob = open(path, "rb").read()
        names = set(m.decode() for m in re.findall(rb"(?<![A-Za-z0-9_])TR_[A-Z][A-Z0-9_]+(?![A-Za-z0-9_])", blob))
        return {n for n in names if not n.startswith(("TR_ERR", "TR_OUT", "TR_FLAG", "TR_LAYOUT", "TR_OK"))}

    release = switches(_ffi.LIB_PATH if "diag" not in os.path.basename(_ffi.LIB_PATH) else os.path.join(ROOT, "tinyrenderer_amd", "libtinyrenderer_hip.so"))
    assert release <= documented, f"undocumented switches in the release library: {sorted(release - documented)}"
    diag = os.path.join(ROOT, "tinyrenderer_amd", "libtinyrenderer_hip_diag.so")
    if os.path.exists(diag):
        extra = switches(diag) - documented
        assert {"TR_RASTER_DEBUG", "TR_DEBUG_PREFIX_STALL", "TR_SHADE_TRACE"} <= extra
