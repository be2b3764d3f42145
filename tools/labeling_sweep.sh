# d = 256: does cutting windows by slots matter at 1-KB rows?  and the clustered generator with uniform windows
for t in "window_balance=-1" "window_balance=25"; do
  timeout -k 10 400 python tools/labeling_experiment.py --d 256 --cases ss,sd,dd --tune "$t" 2>&1 | grep "^\["
done
timeout -k 10 300 python tools/labeling_experiment.py --d 64 --cases ss --graph clustered --tune "window_balance=-1" 2>&1 | grep "^\["
timeout -k 10 300 python tools/labeling_experiment.py --d 64 --cases ss --graph clustered --tune "window_balance=25" 2>&1 | grep "^\["
timeout -k 10 300 python tools/labeling_experiment.py --d 64 --cases ss --graph clustered --tune "sweep=0,walk=0" 2>&1 | grep "^\["
