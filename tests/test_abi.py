"""CPU tests of the C-ABI boundary: the library loads and exports every symbol that
include/anirec.h declares; the ctypes mirrors of the ABI structs have the C layout."""
import os
import re
import subprocess

import pytest

from anime_recommendations_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build(verbose=False)
    return _lib.load()


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "anirec.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(anirec_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libanirec.so does not export %s" % n
        assert n in _lib.PROTOTYPES, "no ctypes prototype for %s" % n
    assert set(_lib.PROTOTYPES) == set(names)


def test_abi_version_and_status_strings(lib):
    assert lib.anirec_abi_version() == _lib.ABI_VERSION
    assert lib.anirec_status_string(0) == b"ok"
    assert b"invalid" in lib.anirec_status_string(-1)


def test_struct_layouts_match_c(tmp_path):
    # compile a tiny C program against the header and compare sizeof/offsetof
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "anirec.h"
int main(void){
  printf("%zu %zu %zu %zu\n", sizeof(anirec_step), sizeof(anirec_state), sizeof(anirec_train_desc), sizeof(anirec_head));
  printf("%zu %zu %zu %zu\n", offsetof(anirec_state, step_fwd), offsetof(anirec_state, loss_wsum), offsetof(anirec_state, val_n), offsetof(anirec_state, reg_anime_sumsq));
  printf("%zu %zu %zu %zu\n", offsetof(anirec_train_desc, W), offsetof(anirec_train_desc, sched), offsetof(anirec_train_desc, packets), offsetof(anirec_train_desc, workspace_bytes));
  return 0; }
'''
    c = tmp_path / "t.c"
    c.write_text(prog)
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    out = [int(x) for x in out]
    import ctypes as C
    S = _lib.STATE_DTYPE
    assert out[0] == _lib.STEP_DTYPE.itemsize == C.sizeof(_lib.Step)
    assert out[1] == S.itemsize
    assert out[2] == C.sizeof(_lib.TrainDesc)
    assert out[3] == C.sizeof(_lib.Head)
    assert out[4:8] == [S.fields["step_fwd"][1], S.fields["loss_wsum"][1], S.fields["val_n"][1],
                        S.fields["reg_anime_sumsq"][1]]
    D = _lib.TrainDesc
    assert out[8:12] == [D.W.offset, D.sched.offset, D.packets.offset, D.workspace_bytes.offset]


def test_size_queries_need_no_gpu(lib):
    assert lib.anirec_packet_floats(10000) == 20004 and lib.anirec_packet_floats(10001) == 20012
    assert lib.anirec_train_workspace_bytes(10000, 8) > 10000 * 128 * 4
    assert lib.anirec_train_workspace_bytes(_lib.MAX_BATCH + 1, 8) == 0
    assert lib.anirec_topk_workspace_bytes(1000, 4) >= 4 * 1000 * 4


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "anime_recommendations_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                assert "import oracle" not in s and "from oracle" not in s, f
