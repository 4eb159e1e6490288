"""MI355X-native batched Overcooked stepper (drop-in for the gym_cooking /
gym_comm hot path of kyle-he/gym-comm).  See DESIGN.md."""
__version__ = "0.1.0"
