"""Training of the feedback GNN (second stage), the loop of /root/reference examples/Feedback_GNN.ipynb cells 2 and 8.

The reference: ``tf.keras.optimizers.Adam(2e-4)``, batches of 100 failed-decoding error patterns, first stage (64 BP
iterations, no gradient) → ``tf.GradientTape`` around ``Second_Stage_GNN_BP_Model`` → gradients clipped to [-10, 10] →
``optimizer.apply_gradients``.  Here the tape is replaced by ``model_stage_two.value_and_grad`` (hand-written reverse
kernels, feedback_gnn_amd/csrc/fgnn_backward.hip); the optimizer arithmetic on the 3 923 parameters is host-framework
tensor work.

    opt = Adam(learning_rate=2e-4)
    for x, z in batches:
        h_vn, lx, lz = model_stage_one(x, z)
        s_hat, b_hat, loss, grads = model_stage_two.value_and_grad(x, z, h_vn, lx, lz)
        grads = [clip_by_value(g, -10, 10) for g in grads]
        opt.apply_gradients(zip(grads, model_stage_two.trainable_weights))
"""
import numpy as np
import torch

__all__ = ["Adam", "clip_by_value", "compute_bler", "harvest_failures", "train_second_stage"]


def clip_by_value(t, lo, hi):
    return torch.clamp(t, lo, hi)


def compute_bler(b, b_hat):
    """Fraction of rows in which ``b`` and ``b_hat`` differ anywhere (sionna/utils/metrics.py compute_bler)."""
    b, b_hat = torch.as_tensor(b), torch.as_tensor(b_hat)
    return float((b != b_hat.to(b.device)).flatten(1).any(1).to(torch.float32).mean())


class Adam:
    """Keras Adam (defaults beta_1=0.9, beta_2=0.999, epsilon=1e-7, no amsgrad):
    ``w -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps)``; ``learning_rate`` may be a callable of the step count."""

    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0
        self._slots = {}

    def apply_gradients(self, grads_and_vars):
        self.iterations += 1
        t = self.iterations
        lr = self.learning_rate(t - 1) if callable(self.learning_rate) else self.learning_rate
        lr_t = float(lr) * np.sqrt(1.0 - self.beta_2 ** t) / (1.0 - self.beta_1 ** t)
        with torch.no_grad():
            for g, v in grads_and_vars:
                if g is None:
                    continue
                m, s = self._slots.setdefault(id(v), (torch.zeros_like(v), torch.zeros_like(v)))
                g = g.to(v.dtype)
                m.mul_(self.beta_1).add_(g, alpha=1.0 - self.beta_1)
                s.mul_(self.beta_2).addcmul_(g, g, value=1.0 - self.beta_2)
                v.sub_(lr_t * m / (s.sqrt() + self.epsilon))


def harvest_failures(model_eval, batch_size, p, count, max_batches=1000, on_device=False):
    """Error patterns the evaluation model fails on (examples/Generate_dataset.ipynb): ``(noise_x, noise_z)`` uint8 arrays
    with at most ``count`` rows, collected with ``Sandwich_BP_GNN_Evaluation_Model.failures`` over at most ``max_batches``
    batches (``p`` is the error weight when the model was built with ``wt=True``)."""
    xs, zs, have = [], [], 0
    for _ in range(max_batches):
        fx, fz = model_eval.failures(batch_size, p)[:2]
        xs.append(fx if on_device else fx.cpu())
        zs.append(fz if on_device else fz.cpu())
        have += int(xs[-1].shape[0])
        if have >= count:
            break
    X, Z = torch.cat(xs)[:count], torch.cat(zs)[:count]
    return (X, Z) if on_device else (X.numpy(), Z.numpy())


def train_second_stage(model_stage_one, model_stage_two, dataset_x, dataset_z, batch_size=100, learning_rate=2e-4,
                       clip_value_grad=10.0, epochs=1, seed=0, log_every=500, log=print, optimizer=None):
    """The training loop of Feedback_GNN.ipynb cell 8 over arrays ``dataset_x``, ``dataset_z`` [N, n] of error patterns.
    Returns the per-step history ``[(loss, bler, flagged_bler), ...]``; updates ``model_stage_two.feedback`` in place."""
    opt = optimizer if optimizer is not None else Adam(learning_rate)
    N = int(dataset_x.shape[0])
    on_device = torch.is_tensor(dataset_x)  # datasets may stay in HBM (uint8 [N, n] tensors) or come as NumPy arrays
    rng = np.random.RandomState(seed)
    history, it = [], 0
    steps = epochs * ((N + batch_size - 1) // batch_size)
    for _ in range(epochs):
        order = rng.permutation(N)  # dataset.shuffle(dataset_size, reshuffle_each_iteration=True)
        for lo in range(0, N, batch_size):
            idx = order[lo:lo + batch_size]
            if on_device:
                idx = torch.from_numpy(idx).to(dataset_x.device)
            x, z = dataset_x[idx], dataset_z[idx]
            it += 1
            h_vn, lx, lz = model_stage_one(x, z)
            s_hat, b_hat, loss, grads = model_stage_two.value_and_grad(x, z, h_vn, lx, lz)
            flagged = compute_bler(torch.zeros_like(s_hat), s_hat)
            bler = compute_bler(torch.zeros_like(b_hat), b_hat)
            history.append((float(loss), bler, flagged))
            if log_every and it % log_every == 0:
                log(f"Iteration {it}/{steps}. Current loss: {float(loss):3f} bler: {bler:.4f} flagged bler: {flagged:.4f}")
            grads = [clip_by_value(g, -clip_value_grad, clip_value_grad) for g in grads]
            opt.apply_gradients(zip(grads, model_stage_two.trainable_weights))
    return history
