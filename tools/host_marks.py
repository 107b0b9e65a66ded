# development tool: host-side time between the prover's marks for ONE warm lone proof (stderr lines "[host] <mark> <us>").
#   python tools/host_marks.py [fib_n]
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from cairo_m_amd.lib import Backend, synth_fibonacci
be = Backend(0)
dev = be.upload_input(synth_fibonacci(int(sys.argv[1]) if len(sys.argv) > 1 else 419000))
for _ in range(4):
    be.prove_device(dev).free()
os.environ["CM_HOST_MARKS"] = "1"
sys.stderr.write("---- marked proof ----\n")
be.prove_device(dev).free()
