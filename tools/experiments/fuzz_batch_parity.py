"""Long soak: tests/fuzzers.py::fuzz_batch_parity from the command line.  usage: fuzz_batch_parity.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzzers
sys.exit(fuzzers.main("batch_parity", sys.argv))
