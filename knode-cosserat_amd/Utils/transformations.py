"""``quaternion_to_euler`` used by the KNODE training loss; drop-in for
``knode_cosserat/Utils/transformations.py`` (reference lines 3-31).  The angle
formulas are the reference's own (they are not the textbook ZYX ones) and are
also what the fused HIP loss kernel implements (``kr_loss_fwd_bwd``)."""
import torch


def quaternion_to_euler(quaternions):
    """[4, a] (w, x, y, z rows) -> [3, a] (roll, pitch, yaw rows), float32."""
    q = quaternions.float()
    q = q / q.norm(p=2, dim=0, keepdim=True)
    w, x, y, z = q.unbind(0)
    roll = torch.atan2(2 * (w * y + x * z), 1 - 2 * (y * y + z * z))
    pitch = torch.asin((2 * (w * z - x * y)).clamp(-1.0, 1.0))
    yaw = torch.atan2(2 * (w * x + y * z), 1 - 2 * (x * x + z * z))
    return torch.stack((roll, pitch, yaw), dim=0)
