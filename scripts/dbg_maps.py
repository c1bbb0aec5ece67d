"""Debug aid: runs bench.py in-process while a daemon thread keeps a copy of /proc/self/maps under gpurun_out/, so that the
raw addresses of a native crash report (e.g. from rocprofv3's signal handler) can be resolved against the libraries afterwards.
Usage: python scripts/dbg_maps.py <bench args>"""
import os
import runpy
import shutil
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", os.environ.get("MAPS_NAME", "maps.txt"))


def _dump():
    while True:
        time.sleep(1.0)
        try:
            shutil.copy("/proc/self/maps", OUT + ".tmp")
            os.replace(OUT + ".tmp", OUT)
        except OSError:
            pass


os.makedirs(os.path.dirname(OUT), exist_ok=True)
threading.Thread(target=_dump, daemon=True).start()
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
