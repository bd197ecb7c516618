#!/usr/bin/env python3
"""Training CLI with the reference's flag surface (train.py:83-107), running the MI355X-native train step.

  python train.py --path DATA --save_path OUT --model_name NAME --opt adam --lr 1e-4 --norm [--mjsynth ...]
  torchrun --nproc-per-node 8 --master-addr 127.0.0.1 train.py ...      # data parallel: one rank per GPU

Differences from the reference, on purpose: `--GRU` / `--norm` mean what they say (the reference's
`from utils import *` shadows both, SURVEY F3: it always builds GRU and always normalises); `--G` selects the
HIP device(s) via HIP_VISIBLE_DEVICES; under torchrun every rank trains on its own (equal-sized) shard of the file list,
starts from rank 0's weights (broadcast), averages gradients with RCCL all-reduces each step, monitors the global
batch-mean loss (so early stopping agrees across ranks) and averages the BatchNorm moving statistics at every epoch end.
"""
import argparse
import os
import pickle
import re
import time
from shutil import rmtree

import numpy as np
from numpy.random import RandomState


def build_parser():
    parser = argparse.ArgumentParser(description='crnn_ctc_loss')
    parser.add_argument('-p', '--path', type=str, required=True)
    parser.add_argument('--training_fname', type=str, required=False, default=None)
    parser.add_argument('--val_fname', type=str, required=False, default="")
    parser.add_argument('--save_path', type=str, required=True)
    parser.add_argument('--model_name', type=str, required=True)
    parser.add_argument('--pretrained_path', default=None, type=str, required=False)
    parser.add_argument('--nbepochs', type=int, default=20)
    parser.add_argument('--G', type=str, default="1")
    parser.add_argument('--random_state', type=int, default=42)
    parser.add_argument('--train_portion', type=float, default=0.9)
    parser.add_argument('--time_dense_size', type=int, default=128)
    parser.add_argument('--n_units', type=int, default=256)
    parser.add_argument('--batch_size', type=int, default=64)
    parser.add_argument('--opt', type=str, default="sgd")
    parser.add_argument('--lr', type=float, default=0.001)
    parser.add_argument('--early_stopping', type=int, default=0)
    parser.add_argument('--norm', action='store_true')
    parser.add_argument('--mjsynth', action='store_true')
    parser.add_argument('--GRU', action='store_true')
    parser.add_argument('--imgh', type=int, default=100)
    parser.add_argument('--imgW', type=int, default=32)
    parser.add_argument('--workers', type=int, default=0,
                        help='image decoding processes feeding the generator (0 = the reference\'s single-threaded loader)')
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("CRNN_DIST_BACKEND", "nccl")     # "nccl" = RCCL over xGMI; "gloo" only to exercise DP on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    elif args.G not in ("", "-1"):
        os.environ.setdefault("HIP_VISIBLE_DEVICES", args.G)

    import utils as U
    from crnn_mi355x.parallel import shard

    out_dir = args.save_path + "/" + args.model_name
    if rank == 0:
        rmtree(out_dir, ignore_errors=True)
        os.makedirs(out_dir)
        with open(out_dir + "/arguments.txt", "w") as f:
            f.write(str(args))
    prng = RandomState(args.random_state)
    lexicon = U.get_lexicon()
    classes = {ch: i for i, ch in enumerate(lexicon)}
    if args.mjsynth:
        train = U.parse_mjsynth(args.path, open(os.path.join(args.path, args.training_fname)).readlines())
        prng.shuffle(train)
        val = U.parse_mjsynth(args.path, open(os.path.join(args.path, args.val_fname)).readlines())
    else:
        train = [os.path.join(dp, f) for dp, dn, fs in os.walk(args.path) for f in fs if re.search('png|jpeg|jpg', f)]
        prng.shuffle(train)
        cut = int(len(train) * args.train_portion)
        train, val = train[:cut], train[cut:]
    max_len = max(U.get_lengths(train).values())
    print(' [INFO] %d train and %d validation images loaded ' % (len(train), len(val)))
    lo, hi = shard(len(train), rank, world)    # equal shards: every rank runs the same number of steps (= collectives) per epoch
    train = train[lo:hi]

    reader = U.Readf(img_size=(args.imgh, args.imgW, 1), normed=args.norm, batch_size=args.batch_size, classes=classes,
                     max_len=max_len, transform_p=0.7, workers=args.workers, seed=args.random_state)
    print(" [INFO] Number of classes: {}; Max. string length: {} ".format(len(classes) + 1, max_len))
    init_model = U.CRNN(num_classes=len(classes) + 1, shape=(args.imgh, args.imgW, 1), GRU=args.GRU,
                        time_dense_size=args.time_dense_size, n_units=args.n_units, max_string_len=max_len)
    model = init_model.get_model()
    if rank == 0:
        U.save_model_json(model, args.save_path, args.model_name)
    if args.pretrained_path is not None:
        model.load_weights(args.pretrained_path)
    train_steps = -(-len(train) // args.batch_size)
    test_steps = -(-len(val) // args.batch_size)
    start_time = time.time()
    if rank == 0:
        with open(out_dir + '/model_summary.txt', 'w') as f:
            model.summary(print_fn=lambda line: f.write(line + '\n'))
        model.summary()
    if args.opt == "adam":
        optimizer = U.optimizers.Adam(lr=args.lr, beta_1=0.5, beta_2=0.999, clipnorm=5)
    else:
        optimizer = U.optimizers.SGD(lr=args.lr, decay=1e-6, momentum=0.9, nesterov=True, clipnorm=5)
    model.compile(loss={"ctc": lambda y_true, y_pred: y_pred}, optimizer=optimizer)
    callbacks_list = []
    if rank == 0:
        callbacks_list.append(U.ModelCheckpoint(filepath=out_dir + '/checkpoint_weights.h5', verbose=1, save_best_only=True, save_weights_only=True))
    if args.early_stopping:
        callbacks_list.append(U.EarlyStoppingIter(monitor='loss', min_delta=.0001, patience=args.early_stopping, verbose=1,
                                                  restore_best_weights=True, mode="auto"))
    down = 2 ** init_model.pooling_counter_h
    H = model.fit_generator(generator=reader.run_generator(train, downsample_factor=down), steps_per_epoch=train_steps,
                            epochs=args.nbepochs, validation_data=reader.run_generator(val, downsample_factor=down) if val else None,
                            validation_steps=test_steps, shuffle=False, verbose=1 if rank == 0 else 0, callbacks=callbacks_list)
    if rank == 0:
        pickle.dump(H.history, open(out_dir + '/loss_history.pickle.dat', 'wb'))
        print(" [INFO] Training finished in %i sec.!" % (round(time.time() - start_time, 2)))
        model.save_weights(out_dir + "/final_weights.h5")
        model.save(out_dir + "/final_model.h5")
        print(" [INFO] Models and history saved! ")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
