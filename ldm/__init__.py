"""`ldm` -- the reference's module surface (python -m ldm.main / ldm.eval_bpd, Experiment, TrainState, VDM...)
re-exported from the MI355X-native implementation in `mulan_amd`."""
