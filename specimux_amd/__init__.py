"""specimux_amd: MI355X-native drop-in for the specimux demultiplexing hot path.

Public names mirror the reference package (src/specimux/__init__.py:14-25).  The compute path is the
HIP library specimux_amd/libsmx.so (C ABI: include/smx.h); nothing here computes alignments on the CPU."""
__version__ = "0.7.0"

from .databases import PrimerDatabase, Specimens  # noqa: E402
from .models import MatchParameters  # noqa: E402
from .io_utils import read_primers_file, read_specimen_file  # noqa: E402
from .orchestration import setup_match_parameters, specimux, specimux_mp  # noqa: E402


def process_sequences(*args, **kwargs):
    """demultiplex.process_sequences (imported lazily so that parsing-only users do not need libsmx.so)."""
    from .demultiplex import process_sequences as impl
    return impl(*args, **kwargs)


__all__ = ["PrimerDatabase", "Specimens", "MatchParameters", "specimux", "specimux_mp", "read_primers_file",
           "read_specimen_file", "setup_match_parameters", "process_sequences"]
