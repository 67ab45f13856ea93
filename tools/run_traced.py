"""Run a script with a watchdog that dumps every thread's Python stack and exits if it is still
running after N seconds:  python tools/run_traced.py N script.py [args...]  (debugging hangs on the GPU box)."""
import faulthandler
import runpy
import sys

secs = int(sys.argv[1])
faulthandler.dump_traceback_later(secs, exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
