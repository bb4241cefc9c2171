"""ocn_amd — MI355X-native OCN common-neighbour predictor hot path."""
