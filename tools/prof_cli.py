import cProfile, pstats, os, sys, shutil, tempfile, io
sys.path.insert(0, os.getcwd())
from specimux_amd import synth, cli
d = tempfile.mkdtemp(prefix="smx_prof_", dir="/dev/shm")
pan = synth.panel_c2(2002)
pf, sf = pan.write(d)
rs = synth.make_reads(pan, 765000, 2002, workers=16)
fq = os.path.join(d, "reads.fastq")
rs.write_fastq_rebuilt(fq, 80, 2002)
out = os.path.join(d, "out")
for rep in range(3):
    shutil.rmtree(out, ignore_errors=True)
    if rep == 2:
        pr = cProfile.Profile(); pr.enable()
    cli.main(["specimux", pf, sf, fq, "-F", "-O", out])
    if rep == 2:
        pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
shutil.rmtree(d, ignore_errors=True)
