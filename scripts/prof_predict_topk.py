import os, sys, runpy
sys.argv = ["time_predict_topk.py"]
runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_predict_topk.py"), run_name="__main__")
