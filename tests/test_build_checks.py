"""The ISA checks build.sh runs on the recorder's hand-written stores (tools/check_asm_stores.py): host logic, no GPU.
hipcc's hazard recogniser does not look into inline asm, so these two rules are what keeps `DFN_GSTORE` (dfn_mlp.h) safe."""
import importlib.util, os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_asm_stores", os.path.join(ROOT, "tools", "check_asm_stores.py"))
cas = importlib.util.module_from_spec(spec)
spec.loader.exec_module(cas)

OK = """
\tv_readfirstlane_b32 s4, v1
\tv_readfirstlane_b32 s5, v2
\tv_mfma_f32_32x32x2_f32 a[0:15], v3, v4, a[0:15]
\ts_add_u32 s4, s4, 0x1000
\ts_addc_u32 s5, s5, 0
\t;;#ASMSTART
\tglobal_store_dword v46, v51, s[4:5] offset:0x500 nt
\t;;#ASMEND
\t;;#ASMSTART
\tglobal_store_dwordx4 v134, v[28:31], s[4:5] offset:0 nt
\ts_nop 1
\t;;#ASMEND
"""
FRESH_BASE = """
\tv_readfirstlane_b32 s4, v1
\tv_readfirstlane_b32 s5, v2
\tv_mov_b32_e32 v9, 0
\t;;#ASMSTART
\tglobal_store_dword v46, v51, s[4:5] offset:0 nt
\t;;#ASMEND
"""
FAR_BASE = FRESH_BASE.replace("\tv_mov_b32_e32 v9, 0\n", "\ts_nop 4\n")          # five wait states: far enough
WIDE_WITHOUT_NOP = """
\ts_mov_b64 s[4:5], s[8:9]
\t;;#ASMSTART
\tglobal_store_dwordx4 v134, v[28:31], s[4:5] offset:0 nt
\ts_nop 0
\t;;#ASMEND
"""
COMPILER_STORE = """
\tv_readlane_b32 s5, v253, 37
\tglobal_store_dwordx2 v0, v[2:3], s[4:5]
"""


def run(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    return cas.main([str(p)])


def test_clean_stores_pass(tmp_path):
    assert run(tmp_path, OK) == 0
    assert run(tmp_path, FAR_BASE) == 0
    assert run(tmp_path, COMPILER_STORE) == 0          # hipcc's own stores are hipcc's business: only inline asm is checked


def test_a_base_fresh_from_a_vector_instruction_is_reported(tmp_path):
    assert run(tmp_path, FRESH_BASE) == 1


def test_a_wide_store_without_its_two_wait_states_is_reported(tmp_path):
    assert run(tmp_path, WIDE_WITHOUT_NOP) == 1
