"""MI355X-native hot path of Atmospheric Athena (hydro CTU integrator + plane-parallel
ionizing radiation).  Host-side mirror of the reference's interface for that path; the
arithmetic lives in csrc/ (HIP, gfx950) behind the C-ABI declared in include/athena_amd.h."""
from . import athinput, config, restart  # noqa: F401
