#!/usr/bin/env python
"""neural_network component — drop-in for the reference's training step
(neural_network/neural_network.py): same entry point, same flags (all strings, booleans via
strtobool), same output artefacts (weights file, model file, History CSV/JSON), with the
train step running on hand-written gfx950 kernels (libanirec) instead of TensorFlow.

Artifacts resolve through the local store (anime_recommendations_amd.artifacts) because
Weights & Biases is unreachable offline; ``--project_name`` and the artifact *type* flags are
accepted and recorded as metadata only.
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from anime_recommendations_amd import artifacts, components as C  # noqa: E402

STR_FLAGS = ["test_size", "embedding_size", "kernel_initializer", "activation_function", "model_loss",
             "optimizer", "start_lr", "min_lr", "max_lr", "batch_size", "rampup_epochs", "sustain_epochs",
             "exp_decay", "weights_artifact", "checkpoint_metric", "save_freq", "mode", "verbose", "epochs",
             "model_name", "input_data", "project_name", "model_artifact", "history_csv", "ID_emb_name",
             "anime_emb_name", "merged_name", "main_df_type", "model_type", "history_type", "weights_type",
             "model_metrics", "l2_reg_factor"]
BOOL_FLAGS = ["TPU_INIT", "save_weights_only", "save_best_weights", "save_model"]

logger = C.setup_logging("neural_network")


def go(args):
    from anime_recommendations_amd import data, ingest, trainer, weights_io
    # the graph is fixed by the kernels: reject configurations they do not implement
    for flag, want in (("model_loss", "binary_crossentropy"), ("optimizer", "adam"),
                       ("activation_function", "sigmoid"), ("kernel_initializer", "he_normal")):
        if str(getattr(args, flag)).lower() != want:
            raise ValueError("--%s %r is not supported by the HIP train step (only %r)"
                             % (flag, getattr(args, flag), want))
    if args.TPU_INIT:
        logger.info("TPU_INIT requested: ignored, training runs on MI355X (use torchrun for >1 GPU)")
    logger.info("Loading data artifact %s", args.input_data)
    # one process per GPU under torchrun (RANK/WORLD_SIZE set): pick this rank's card before anything
    # allocates on it, so the id tables of get_df land on cuda:LOCAL_RANK and not on cuda:0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", rank)) if world > 1 else 0
    import torch
    if torch.cuda.is_available():
        local %= torch.cuda.device_count()
        torch.cuda.set_device(local)
    # get_df (neural_network.py:25-63 of the reference): the id encoding runs on the GPU (ingest.py)
    table = ingest.load_user_stats(artifacts.use_artifact(args.input_data, args.main_df_type),
                                   device="cuda:%d" % local)
    logger.info("Final df shape is (%d, 3); %d users, %d anime", len(table), table.n_users, table.n_anime)
    cfg = trainer.FitConfig(
        epochs=int(args.epochs), batch_size=int(args.batch_size), test_size=int(args.test_size),
        embedding_size=int(args.embedding_size), l2_reg_factor=float(args.l2_reg_factor),
        start_lr=float(args.start_lr), min_lr=float(args.min_lr), max_lr=float(args.max_lr),
        rampup_epochs=int(args.rampup_epochs), sustain_epochs=int(args.sustain_epochs),
        exp_decay=float(args.exp_decay), monitor=args.checkpoint_metric, mode=args.mode,
        verbose=int(args.verbose), seed=int(os.environ.get("ANIREC_SEED", "0")))
    # >1 rank: ratings sharded by user over RCCL.  The reference's TPU branch (neural_network.py:173-178)
    # computes batch_size * replicas and max_lr * replicas but never uses them: model.fit gets
    # args.batch_size (:213) and lrfn reads args.max_lr (:113), so the GLOBAL batch and the schedule are
    # unchanged by the replica count.  That actual behaviour is the default here (each rank takes
    # batch_size // world of every global batch: same steps per epoch, same History, same early stop as
    # one GPU); ANIREC_WEAK_SCALING=1 selects the branch's apparent intent instead (per-rank batch_size,
    # max_lr * world).
    engine = None
    if world > 1:
        import torch.distributed as dist
        from anime_recommendations_amd.dist import DistTrainEngine
        if not dist.is_initialized():
            dist.init_process_group(os.environ.get("ANIREC_DIST_BACKEND", "nccl"))
        n_train = len(table) - cfg.test_size
        if os.environ.get("ANIREC_WEAK_SCALING") == "1":
            per_rank = cfg.batch_size
            cfg.max_lr = cfg.max_lr * world
        else:
            if cfg.batch_size % world:
                # a remainder would silently shrink the global batch: other steps per epoch, Adam step sizes, History
                raise ValueError("batch_size %d is not a multiple of the %d ranks: the global batch of model.fit "
                                 "(neural_network.py:213) must be cut into equal per-rank parts (or set "
                                 "ANIREC_WEAK_SCALING=1 for batch_size ratings PER rank)" % (cfg.batch_size, world))
            per_rank = cfg.batch_size // world
        engine = DistTrainEngine(table.n_users, table.n_anime, min(per_rank, max(1, n_train // world)),
                                 l2=cfg.l2_reg_factor, device="cuda:%d" % local)
        if rank != 0:
            cfg.verbose = 0
    res = trainer.fit(table, cfg, engine=engine, log=lambda s: (print(s), logger.info(s)))
    logger.info("model trained")
    if rank != 0:
        return res

    def stem(p):
        return os.path.splitext(p)[0] + ".safetensors"

    # ModelCheckpoint(filepath=weights_artifact, save_best_only): the best-val_loss weights
    wpath = stem(args.weights_artifact)
    bU, bA, bh = (res.best_U, res.best_A, res.best_head) if res.best_U is not None else (res.U, res.A, res.head)
    weights_io.save_model(wpath, bU, bA, bh, table.user_ids, table.anime_ids, args.ID_emb_name, args.anime_emb_name)
    mpath = stem(args.model_name)
    if args.save_model:
        weights_io.save_model(mpath, res.U, res.A, res.head, table.user_ids, table.anime_ids,
                              args.ID_emb_name, args.anime_emb_name, optimizer=res.optimizer,
                              extra={"best_epoch": res.best_epoch, "stopped_epoch": res.stopped_epoch})
    hist = trainer.history_frame(res.history)
    with open("history.json", "w") as f:
        hist.to_json(f)
    hist.to_csv(args.history_csv)
    artifacts.log_artifact(args.weights_artifact, wpath, args.weights_type, "file containing all weights")
    artifacts.log_artifact(args.history_csv, args.history_csv, args.history_type,
                           "csv file of neural network training history")
    if args.save_model:
        artifacts.log_artifact(args.model_artifact, mpath, args.model_type, "trained neural network",
                               metadata={"Loss function": args.model_loss, "Optimizer": args.optimizer,
                                         "Activation function": args.activation_function,
                                         "Start learning rate": args.start_lr, "Min learning rate": args.min_lr,
                                         "Max learning rate": args.max_lr, "Batch size": args.batch_size,
                                         "L2 regularization factor": args.l2_reg_factor,
                                         "Monitored metrics": str(args.model_metrics)})
    logger.info("Artifacts logged")
    print(json.dumps({"epochs_run": len(hist), "best_epoch": res.best_epoch,
                      "val_loss": res.history["val_loss"][res.best_epoch] if res.best_epoch >= 0 else None}))
    return res


if __name__ == "__main__":
    _args = C.make_parser("Train an anime recommendation neural network", STR_FLAGS, BOOL_FLAGS).parse_args()
    try:
        go(_args)
    except Exception:                      # non-zero exit + the reason in ./neural_network.log
        logger.exception("neural_network failed")
        raise
