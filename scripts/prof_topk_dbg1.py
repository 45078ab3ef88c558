import os
os.environ["ANIREC_TOPK_DEBUG"] = "1"
import runpy, sys
sys.argv = ["time_topk.py", "350000", "65536"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_topk.py"), run_name="__main__")
