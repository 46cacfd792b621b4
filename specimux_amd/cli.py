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


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Specimux: Demultiplex MinION sequences by dual barcode indexes and primers.")
    p.add_argument("primer_file", help="Fasta file containing primer information")
    p.add_argument("specimen_file", help="TSV file containing specimen mapping with barcodes and primers")
    p.add_argument("sequence_file", help="Sequence file in Fasta or Fastq format, gzipped or plain text")
    p.add_argument("--min-length", type=int, default=-1, help="Minimum sequence length.  Shorter sequences will be skipped (default: no filtering)")
    p.add_argument("--max-length", type=int, default=-1, help="Maximum sequence length.  Longer sequences will be skipped (default: no filtering)")
    p.add_argument("-n", "--num-seqs", type=str, default="-1", help="Number of sequences to read from file (e.g., -n 100 or -n 102,3)")
    p.add_argument("-e", "--index-edit-distance", type=int, default=-1, help="Barcode edit distance value, default is half of min distance between barcodes")
    p.add_argument("-E", "--primer-edit-distance", type=int, default=-1, help="Primer edit distance value, default is min distance between primers")
    p.add_argument("-l", "--search-len", type=int, default=80, help="Length to search for index and primer at start and end of sequence (default: 80)")
    p.add_argument("-F", "--output-to-files", action="store_true", help="Create individual sample files for sequences")
    p.add_argument("-P", "--output-file-prefix", default="", help="Prefix for individual files when using -F (default: no prefix)")
    p.add_argument("-O", "--output-dir", default=".", help="Directory for individual files when using -F (default: .)")
    p.add_argument("--color", action="store_true", help="Highlight barcode matches in blue, primer matches in green")
    p.add_argument("--trim", choices=[TrimMode.NONE, TrimMode.TAILS, TrimMode.BARCODES, TrimMode.PRIMERS], default=TrimMode.BARCODES, help="trimming to apply")
    p.add_argument("--dereplicate", choices=[MultipleMatchStrategy.NONE, MultipleMatchStrategy.BEST], default=MultipleMatchStrategy.BEST,
                   help="Dereplication strategy: 'best' selects best match per specimen/barcode group (default), 'none' outputs all matches")
    p.add_argument("-d", "--diagnostics", nargs="?", const=1, type=int, choices=[1, 2, 3],
                   help="Enable diagnostic trace logging: 1=standard (default), 2=detailed, 3=verbose")
    p.add_argument("-D", "--debug", action="store_true", help="Enable debug logging")
    p.add_argument("--disable-prefilter", action="store_true", help="Disable barcode prefiltering (bloom filter optimization)")
    p.add_argument("--disable-preorient", action="store_true", help="Disable heuristic pre-orientation")
    p.add_argument("-t", "--threads", type=int, default=-1, help="Number of worker threads to use")
    p.add_argument("--sample-topq", type=int, default=0, metavar="N",
                   help="Create subsample directories with top N sequences by average quality score (default: disabled)")
    p.add_argument("-v", "--version", action="version", version=version())
    return p


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
    setup_logging(args.debug, args.output_dir if args.output_to_files else None)
    logging.info(f"Starting {version()}")
    logging.info(f"Command line: {' '.join(argv)}")
    from . import orchestration
    if args.output_to_files:
        orchestration.specimux_mp(args)
    else:
        if args.threads > 1:
            logging.warning(f"Multithreading only supported for file output. Ignoring --threads {args.threads}")
        orchestration.specimux(args)


if __name__ == "__main__":
    main()
