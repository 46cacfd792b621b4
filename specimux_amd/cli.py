"""Command line of the drop-in `specimux` entry point (reference: src/specimux/cli.py:15-110).
Same positional arguments and flags."""
import argparse
import logging
import os
import sys

from . import __version__
from .constants import MultipleMatchStrategy, TrimMode


def version() -> str:
    return f"specimux version {__version__} (specimux_amd, MI355X)"


# The reference's flag surface (src/specimux/cli.py:19-44), kept verbatim so that existing command lines and
# `--help` output carry over: (flags, argparse keyword arguments).
_POSITIONALS = [
    ("primer_file", "Fasta file containing primer information"),
    ("specimen_file", "TSV file containing specimen mapping with barcodes and primers"),
    ("sequence_file", "Sequence file in Fasta or Fastq format, gzipped or plain text"),
]
_INT, _FLAG = dict(type=int), dict(action="store_true")
_OPTIONS = [
    (("--min-length",), dict(_INT, default=-1, help="Minimum sequence length.  Shorter sequences will be skipped (default: no filtering)")),
    (("--max-length",), dict(_INT, default=-1, help="Maximum sequence length.  Longer sequences will be skipped (default: no filtering)")),
    (("-n", "--num-seqs"), dict(type=str, default="-1", help="Number of sequences to read from file (e.g., -n 100 or -n 102,3)")),
    (("-e", "--index-edit-distance"), dict(_INT, default=-1, help="Barcode edit distance value, default is half of min distance between barcodes")),
    (("-E", "--primer-edit-distance"), dict(_INT, default=-1, help="Primer edit distance value, default is min distance between primers")),
    (("-l", "--search-len"), dict(_INT, default=80, help="Length to search for index and primer at start and end of sequence (default: 80)")),
    (("-F", "--output-to-files"), dict(_FLAG, help="Create individual sample files for sequences")),
    (("-P", "--output-file-prefix"), dict(default="", help="Prefix for individual files when using -F (default: no prefix)")),
    (("-O", "--output-dir"), dict(default=".", help="Directory for individual files when using -F (default: .)")),
    (("--color",), dict(_FLAG, help="Highlight barcode matches in blue, primer matches in green")),
    (("--trim",), dict(choices=[TrimMode.NONE, TrimMode.TAILS, TrimMode.BARCODES, TrimMode.PRIMERS], default=TrimMode.BARCODES,
                       help="trimming to apply")),
    (("--dereplicate",), dict(choices=[MultipleMatchStrategy.NONE, MultipleMatchStrategy.BEST], default=MultipleMatchStrategy.BEST,
                              help="Dereplication strategy: 'best' selects best match per specimen/barcode group (default), "
                                   "'none' outputs all matches")),
    (("-d", "--diagnostics"), dict(_INT, nargs="?", const=1, choices=[1, 2, 3],
                                   help="Enable diagnostic trace logging: 1=standard (default), 2=detailed, 3=verbose")),
    (("-D", "--debug"), dict(_FLAG, help="Enable debug logging")),
    (("--disable-prefilter",), dict(_FLAG, help="Disable barcode prefiltering (bloom filter optimization)")),
    (("--disable-preorient",), dict(_FLAG, help="Disable heuristic pre-orientation")),
    (("-t", "--threads"), dict(_INT, default=-1, help="Number of worker threads to use")),
    (("--sample-topq",), dict(_INT, default=0, metavar="N",
                              help="Create subsample directories with top N sequences by average quality score (default: disabled)")),
]


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser(description="Specimux: Demultiplex MinION sequences by dual barcode indexes and primers.")
    for name, text in _POSITIONALS:
        parser.add_argument(name, help=text)
    for flags, kwargs in _OPTIONS:
        parser.add_argument(*flags, **kwargs)
    parser.add_argument("-v", "--version", action="version", version=version())
    return parser


def parse_args(argv):
    parser = build_parser()
    args = parser.parse_args(argv[1:])
    text = args.num_seqs
    try:
        if "," in text:
            first, count = text.split(",")
            args.start_seq, args.num_seqs = int(first), int(count)
        else:
            args.start_seq, args.num_seqs = 1, int(text)
    except ValueError:
        parser.error("Invalid format for -n option. Use an integer or 'start,num' with integers.")
    return args


def setup_logging(debug: bool, output_dir: str = None, is_worker: bool = False):
    root = logging.getLogger()
    root.handlers.clear()
    fmt = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")
    console = logging.StreamHandler()
    console.setFormatter(fmt)
    root.addHandler(console)
    if output_dir and not is_worker:
        os.makedirs(output_dir, exist_ok=True)
        fileh = logging.FileHandler(os.path.join(output_dir, "log.txt"), mode="w")
        fileh.setFormatter(fmt)
        root.addHandler(fileh)
    root.setLevel(logging.DEBUG if debug else logging.INFO)


def main(argv=None):
    argv = sys.argv if argv is None else argv
    args = parse_args(argv)
    # under a one-process-per-GPU launch (torch.distributed.run) only rank 0 owns log.txt (the reference's workers log
    # to the console only: multiprocessing_utils.py)
    setup_logging(args.debug, args.output_dir if args.output_to_files else None, is_worker=int(os.environ.get("RANK", "0")) > 0)
    logging.info(f"Starting {version()}")
    logging.info(f"Command line: {' '.join(argv)}")
    from . import orchestration
    if args.output_to_files and args.threads > 0 and "SMX_IO_THREADS" not in os.environ:
        # -t N (the reference's worker-pool size, orchestration.py:175): here the size of the host I/O pool -- reader, window
        # packer and writer threads -- read by libsmx.so when it is first used; the matching itself needs no host threads
        os.environ["SMX_IO_THREADS"] = str(args.threads)
    if args.output_to_files:
        orchestration.specimux_mp(args)
    else:
        if args.threads > 1:
            logging.warning(f"Multithreading only supported for file output. Ignoring --threads {args.threads}")
        orchestration.specimux(args)


if __name__ == "__main__":
    main()
