"""ctc_amd: MI355X (gfx950) tensor-contraction backend behind a C ABI (include/ctc_tc.h)."""
from .host import (BF16, CLASS_NAMES, F32, F64, LAYOUT_NAMES, CtcError, Plan, contract,  # noqa: F401
                   device_count, fill, get_option, grouped_contract, noalloc_scratch_bytes, pack_small_problems, permute, set_option)
