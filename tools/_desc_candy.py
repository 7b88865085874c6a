import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import shadernn_amd as snn
from test_param_import import _zoo
ctx = snn.Context(0)
net = _zoo("candy-9_simplified-opt", input_shape=(96, 128, 3))
r = snn.GraphRunner(ctx, net, 1, 96, 128, dtype=snn.F16)
for d in r.describe(): print(d[:200])
