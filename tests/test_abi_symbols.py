"""The C-ABI library loads (no GPU needed) and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def libpath():
    from face_vijnana_yolov3_amd.build import build_library
    return build_library()


def declared_symbols():
    syms = set()
    for h in glob.glob(os.path.join(ROOT, 'include', '*.h')):
        txt = open(h).read()
        txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
        syms |= set(re.findall(r'\b(fv_[a-z0-9_]+)\s*\(', txt))
    return sorted(syms)


def test_header_declares_entry_points():
    syms = declared_symbols()
    for s in ('fv_create', 'fv_destroy', 'fv_last_error', 'fv_decode_nms'):
        assert s in syms


def test_library_exports_every_declared_symbol(libpath):
    L = ctypes.CDLL(libpath)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, missing
    L.fv_abi_version.restype = ctypes.c_int
    assert L.fv_abi_version() >= 1


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from face_vijnana_yolov3_amd._lib import Context, FvError
    with pytest.raises(FvError):
        Context(0)


def test_product_does_not_import_oracle():
    """No file of the product package may reference the oracle (it is test infrastructure)."""
    for f in glob.glob(os.path.join(ROOT, 'face_vijnana_yolov3_amd', '**', '*.py'), recursive=True):
        src = open(f).read()
        assert not re.search(r'^\s*(from|import)\s+\.*oracle', src, flags=re.M), f
