"""MI355X-native message-passing core for HyperGraphNets."""
