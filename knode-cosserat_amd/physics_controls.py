"""Tendon-tension generators; drop-in for ``knode_cosserat/physics_controls.py``
(reference lines 3-33).  Host-side input generation only (NumPy), kept
bit-identical to the reference, including its use of the global NumPy RNG."""
import numpy as np

_QUARTER = 2 * np.pi / 4


def calc_controls(control_type, control_arg, del_t, train_len):
    np.random.seed(int(control_arg))  # only the 'random' type draws from it
    controls = []
    for i in range(1, train_len + 1):
        if control_type == 'sine':
            period_steps = control_arg / del_t
            row = [6 + np.sin(2 * np.pi * i / period_steps + k * _QUARTER) for k in range(4)]
        elif control_type == 'step':
            bump = 0 if i * del_t < 1.5 else control_arg
            row = [5 + bump, 5, 5, 5 + bump]
        elif control_type == 'random':
            row = [5 + 5 * np.random.rand() for _ in range(4)]
        else:  # the reference's 'ramp' branch dereferences an undefined name, i.e. it raises as well
            raise Exception('Unknown control type ' + control_type)
        controls.append(row)
    return controls
