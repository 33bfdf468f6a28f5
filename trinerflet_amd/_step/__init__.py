"""The parts of trinerflet_amd.train.TrainStep (mixins, one per concern); train.py holds the constructor and the stages of a step."""
