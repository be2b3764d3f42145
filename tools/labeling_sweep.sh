for t in "walk_drift=3" "walk_drift=1" "walk_drift=2" "walk_steps=4" "walk_steps=1" "walk_steps=4,walk_drift=2"; do
  timeout -k 10 300 python tools/labeling_experiment.py --d 64 --cases sd,ds --tune "$t" 2>&1 | grep "^\["
done
