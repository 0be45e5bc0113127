"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every declared symbol."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _declared_symbols():
    text = (ROOT / "include" / "internnav_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|void|char\s*\*|const char\s*\*)\s*\*?\s*(ina_[a-z0-9_]+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_header_symbols_match_binding(built_lib):
    from internnav_amd import _lib

    declared = _declared_symbols()
    assert declared, "no symbols parsed from include/internnav_amd.h"
    assert sorted(_lib.SYMBOLS) == declared


def test_library_exports_every_symbol(built_lib):
    h = ctypes.CDLL(str(built_lib))
    for name in _declared_symbols():
        assert hasattr(h, name), f"{name} declared in include/internnav_amd.h but not exported"
    from internnav_amd import _lib

    assert _lib.lib().ina_abi_version() == _lib.ABI_VERSION == 8


def test_graft_entry_build_passes_on_the_current_abi(built_lib):
    """__graft_entry__.build() is what the driver runs every round: it must accept the library the tree builds (an ABI bump once left a pinned
    version number behind in it)."""
    import __graft_entry__ as g

    g.build()


def test_struct_sizes_match_header(built_lib):
    """ctypes mirrors must have exactly the compiled C layout (ina_struct_size reports sizeof of each argument struct)."""
    from internnav_amd import _lib

    mirrors = [_lib.GemmArgs, _lib.AttnArgs, _lib.NormArgs, _lib.PatchifyArgs, _lib.Embed3Args, _lib.Head3Args,
               _lib.SeqpoolArgs, _lib.SelectArgs, _lib.PoolActArgs, _lib.GatherArgs,
               _lib.RopeArgs, _lib.MropeTableArgs, _lib.ArgmaxArgs, _lib.DitAttnArgs, _lib.ResizeU8Args, _lib.QwenPatchifyArgs, _lib.U8LutArgs, _lib.ResizeF32Args, _lib.GnMishArgs, _lib.PadRowsArgs, _lib.DdimStepArgs,
               _lib.EwArgs, _lib.ColsumArgs, _lib.NormBwdArgs, _lib.TransposeArgs, _lib.SparseRowsArgs, _lib.SmallLinearArgs, _lib.MseArgs,
               _lib.AdamwArgs, _lib.GemmNnArgs, _lib.AttnBwdArgs, _lib.DitRowchainArgs, _lib.GemmDwArgs]
    for k, m in enumerate(mirrors):
        assert ctypes.sizeof(m) == _lib.lib().ina_struct_size(k), f"struct {k} ({m.__name__}) layout mismatch"
    assert _lib.lib().ina_struct_size(len(mirrors)) == -1


def test_product_package_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under internnav_amd/ may import it (no CPU fallback on the product path)."""
    for f in (ROOT / "internnav_amd").rglob("*.py"):
        src = f.read_text()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
