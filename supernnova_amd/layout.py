"""state_dict <-> canonical flat parameter order.

Key names are the reference's:
  vanilla      rnn_layer.{weight_ih,weight_hh,bias_ih,bias_hh}_l{l}{,_reverse}, output_layer.{weight,bias}
               (training/vanilla_rnn.py:26-34)
  variational  rnn_layer{l}.module.{weight_ih_l0,weight_hh_l0_raw,bias_ih_l0,bias_hh_l0,
               weight_ih_l0_reverse,weight_hh_l0_reverse,...}, output_layer.{weight,bias}
               (training/variational_rnn.py:29-52, 216-222)
  bayesian     rnn_layer.{weight_ih,weight_hh,bias_ih,bias_hh}_{mu,rho}_l{l}{,_reverse},
               output_layer.{mu,rho,bias}   (training/bayesian_rnn.py:247-266, 583-592)
The canonical flat order is torch's _flat_weights order (= the reference's eps
draw order, bayesian_rnn.py:472-528): per (layer, direction) w_ih, w_hh, b_ih,
b_hh; then the head weight and the head bias.
"""


def vanilla_keys(num_layers, dirs):
    keys = []
    for l in range(num_layers):
        for d in range(dirs):
            s = "_reverse" if d else ""
            keys += [f"rnn_layer.{n}_l{l}{s}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    return keys + ["output_layer.weight", "output_layer.bias"]


def variational_keys(num_layers, dirs):
    keys = []
    for l in range(num_layers):
        for d in range(dirs):
            s = "_reverse" if d else ""
            hh = "weight_hh_l0_raw" if d == 0 else "weight_hh_l0_reverse"
            keys += [f"rnn_layer{l}.module.{n}" for n in (f"weight_ih_l0{s}", hh, f"bias_ih_l0{s}", f"bias_hh_l0{s}")]
    return keys + ["output_layer.weight", "output_layer.bias"]


def bayesian_keys(num_layers, dirs, which):
    """which: 'mu' (ends with the deterministic head bias) or 'rho'."""
    keys = []
    for l in range(num_layers):
        for d in range(dirs):
            s = "_reverse" if d else ""
            keys += [f"rnn_layer.{n}_{which}_l{l}{s}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    keys.append(f"output_layer.{which}")
    if which == "mu":
        keys.append("output_layer.bias")
    return keys
