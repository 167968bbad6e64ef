"""Compile-time constants of the kernels that host-side tools need, parsed from the headers (one source of truth:
the .h files under csrc/)."""
import os
import re

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def sweep2_infl(nq):
    """Cfg<NQ>::INFL of gso_sweep2.h: LDS-DMA entries a wave of the sweep kernel keeps in flight."""
    src = open(os.path.join(_CSRC, "gso_sweep2.h")).read()
    lds3 = int(re.search(r"#define FPHIP_S2_NQ3_LDS (\d+)", src).group(1))
    wave_lds = 13312 if nq == 4 else lds3 if nq == 3 else 9984
    esz = nq * 256
    npair = min(16, wave_lds // (2 * esz))
    return 2 * (npair - 1)
