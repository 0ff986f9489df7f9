"""Model / step configuration: the defaults of /root/reference/src/args.py that define the published
model, readable from the reference's own argparse namespace."""


class Config:
    num_queries = 100
    hidden_dim = 256
    nheads = 8
    enc_layers = 6
    dec_layers = 6
    dim_feedforward = 2048
    dropout = 0.1
    num_ref = 20
    dense_trans_dim = 512
    dense_trans_layers = (4,)
    class_trans_layers = (2, 2, 1)
    class_token_dim = 64
    depth_interval = (0.1, 0.3, 0.5, 0.7, 0.9)
    interval_sample_num = (30, 80, 160)
    min_depth_eval = 1e-3
    max_depth_eval = 10.0
    max_depth = 10
    depth_loss_weights = (0.25, 0.25, 0.25, 1.0)
    seg_loss_weight = 2.0
    variance_focus = 0.85
    log_depth_error = False          # argparse default; the published run passes --log_depth_error
    set_cost_class = 1.0
    set_cost_line = 5.0
    line_loss_coef = 5.0
    eos_coef = 0.1
    aux_loss = True
    lr = 1e-4
    lr_backbone = 1e-5
    weight_decay = 1e-4
    clip_max_norm = 0.1
    device = "cuda"
    with_line = True
    with_center = True
    with_dense = True
    with_plane_norm_loss = False
    plane_norm_loss_coef = 50.0          # src/args.py:81

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @classmethod
    def from_args(cls, args):
        """Accepts the reference's argparse.Namespace (src/args.py) or a Config."""
        if isinstance(args, cls):
            return args
        cfg = cls()
        for k in dir(cls):
            if k.startswith("_") or callable(getattr(cls, k)):
                continue
            if hasattr(args, k):
                v = getattr(args, k)
                setattr(cfg, k, tuple(v) if isinstance(v, list) else v)
        if not (cfg.with_line and cfg.with_center and cfg.with_dense):
            # the only flag combination of the reference that constructs (header)
            raise ValueError("gw_depth_amd implements the --with_line --with_center --with_dense model")
        if getattr(args, "with_line_depth", False) or getattr(args, "with_dense_center", False):
            raise ValueError("--with_line_depth / --with_dense_center are not on the accelerated path")
        return cfg
