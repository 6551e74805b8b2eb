"""Checkpoint interchange with the reference (SURVEY.md §5 checkpoint row, §8f rank 3).

The reference saves `model.state_dict()` through boilr (absent). The key scheme and tensor shapes of the HIP model are
identical, but its parameters are strided views of one flat arena; `state_dict_reference_layout` returns plain contiguous
CPU tensors in the reference's memory layout, so a file written here loads into the reference's `LadderVAE` with
`load_state_dict` and vice versa. Optimiser state is stored per parameter name (torch.optim.Adamax names: exp_avg,
exp_inf) so it can be re-attached to either implementation.
"""
import torch


def state_dict_reference_layout(model):
    """{key: contiguous CPU tensor} with the reference's keys, shapes and (row-major) layout."""
    return {k: v.detach().to('cpu').contiguous().clone() for k, v in model.state_dict().items()}


def optimizer_state_by_name(model, optimizer):
    """{'step': int, 'state': {param_name: {'exp_avg', 'exp_inf'}}} in the parameters' logical shapes."""
    arena = model.pack()
    optimizer._state()
    out = {}
    named = dict(model.named_parameters())
    for name, (off, n) in arena.slots.items():
        p = named[name]
        if not p.requires_grad:
            continue
        # same strides as the parameter view -> same logical element order
        m = torch.as_strided(optimizer.exp_avg, p.shape, p.stride(), off).detach().cpu().contiguous().clone()
        u = torch.as_strided(optimizer.exp_inf, p.shape, p.stride(), off).detach().cpu().contiguous().clone()
        out[name] = {'exp_avg': m, 'exp_inf': u}
    return {'step': int(optimizer.step_count.item()), 'state': out, 'lr': optimizer.lr, 'betas': optimizer.betas,
            'eps': optimizer.eps, 'weight_decay': optimizer.weight_decay}


def save_checkpoint(path, model, optimizer=None):
    ck = {'model': state_dict_reference_layout(model), 'global_step': int(model.global_step)}
    if optimizer is not None:
        ck['optimizer'] = optimizer_state_by_name(model, optimizer)
    torch.save(ck, path)


def load_checkpoint(path, model, optimizer=None):
    """Loads a file written by `save_checkpoint`, or a bare reference `state_dict` file."""
    ck = torch.load(path, map_location='cpu')
    sd = ck['model'] if isinstance(ck, dict) and 'model' in ck else ck
    model.load_state_dict(sd)
    if isinstance(ck, dict):
        model.global_step = int(ck.get('global_step', model.global_step))
    if optimizer is not None and isinstance(ck, dict) and 'optimizer' in ck:
        arena = model.pack()
        optimizer._state()
        named = dict(model.named_parameters())
        for name, st in ck['optimizer']['state'].items():
            off, n = arena.slots[name]
            p = named[name]
            torch.as_strided(optimizer.exp_avg, p.shape, p.stride(), off).copy_(st['exp_avg'])
            torch.as_strided(optimizer.exp_inf, p.shape, p.stride(), off).copy_(st['exp_inf'])
        optimizer.step_count.fill_(int(ck['optimizer']['step']))
    return ck
