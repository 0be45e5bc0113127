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

    assert _lib.lib().ina_abi_version() == 1


def test_struct_sizes_match_header(built_lib):
    """ctypes mirrors must have the C layout (checked against sizes the library reports)."""
    from internnav_amd import _lib

    assert ctypes.sizeof(_lib.GemmArgs) == 7 * 8 + 13 * 4 + 4 * 8 + 2 * 4 + (4 if (7 * 8 + 13 * 4) % 8 else 0)
    assert ctypes.sizeof(_lib.AttnArgs) % 8 == 0
    assert ctypes.sizeof(_lib.NormArgs) % 8 == 0
