"""digdriver_amd -- MI355X-native implementation of DIGDriver's mutation-rate + burden-test
hot path (hand-written HIP kernels behind a C ABI; Python host mirroring the reference's
function names).  See DESIGN.md for scope and INTEGRATION.md for the drop-in boundary."""
__version__ = "0.1.0"
