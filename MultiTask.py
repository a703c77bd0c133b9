#!/usr/bin/env python3
"""Sweep launcher — the counterpart of the reference's MultiTask.py (MultiTask.py:1-125): a YAML with a `Static` option
tree and a `Dynamic` list of PRODUCT / CONCAT combinators over dotted overrides expands into one SingleTask YAML per
combination; every task is one `python main.py -p <task>.yaml` process.

    python MultiTask.py -p opt/MultiTask/default.yaml -g 0,1,2,3

The reference farms the tasks over NVIDIA GPUs picked by free memory (utils/TasksManager.py, pynvml); here a task takes
one whole GPU from the -g list (HIP_VISIBLE_DEVICES) and at most len(-g) (or -m) tasks run at a time; a task that exits
non-zero is reported and NOT re-queued forever (the reference re-pends failed tasks without bound)."""
import argparse
import itertools
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import yaml  # noqa: E402


def tree_to_dotlist(tree, prefix=""):
    """omegaconf2dotlist (utils/misc.py:53-54): 'a.b.c=value' for every leaf (lists are leaves)"""
    out = []
    for k, v in tree.items():
        key = prefix + str(k)
        if isinstance(v, dict):
            out.extend(tree_to_dotlist(v, key + "."))
        else:
            out.append((key, v))
    return out


def expand(spec):
    """dict2dotlist_list / PRODUCT / CONCAT (MultiTask.py:27-56): a list of override lists [(dotted key, value), ...]"""
    if "PRODUCT" in spec:
        parts = [expand(s) for s in spec["PRODUCT"]]
        return [sum(combo, []) for combo in itertools.product(*parts)]
    if "CONCAT" in spec:
        return [dl for s in spec["CONCAT"] for dl in expand(s)]
    return [[(k, v) for k, v in spec.items()]]


def dotlist_to_tree(pairs):
    """OmegaConf.from_dotlist: later assignments override earlier ones"""
    tree = {}
    for key, value in pairs:
        node = tree
        parts = key.split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = value
    return tree


def gen_task_list(yaml_path, main_script_path):
    """MultiTask.py:62-85: (task name, command, yaml path) per combination; task YAMLs go to temp_opt_<project> next to
    the sweep YAML"""
    with open(yaml_path) as f:
        opt = yaml.safe_load(f)
    temp_dir = os.path.join(os.path.dirname(os.path.abspath(yaml_path)), "temp_opt_" + str(opt["Static"]["Log"]["project_name"]))
    os.makedirs(temp_dir, exist_ok=True)
    static = tree_to_dotlist(opt["Static"])
    tasks = []
    for idx, dynamic in enumerate(expand({"CONCAT": opt["Dynamic"]})):
        tree = dotlist_to_tree(static + dynamic)
        tree.pop("Source", None)                       # scheduler hints of the reference's farm (gpucost / cpucost)
        name = "exp_{:03}".format(idx)
        path = os.path.join(temp_dir, name + ".yaml")
        with open(path, "w") as f:
            yaml.safe_dump(tree, f, sort_keys=False)
        tasks.append((name, [sys.executable, main_script_path, "-p", path], path))
    return tasks, temp_dir


def run(tasks, gpus, max_task, interval):
    free, running, failed = list(gpus), [], []
    pending = list(tasks)
    while pending or running:
        while pending and free and len(running) < max_task:
            name, cmd, _ = pending.pop(0)
            g = free.pop(0)
            env = dict(os.environ, HIP_VISIBLE_DEVICES=str(g))
            print("[MultiTask] %s on GPU %s: %s" % (name, g, " ".join(cmd)), flush=True)
            running.append((name, g, subprocess.Popen(cmd + ["-g", "0"], env=env)))
        still = []
        for name, g, proc in running:
            rc = proc.poll()
            if rc is None:
                still.append((name, g, proc))
                continue
            free.append(g)
            if rc != 0:
                failed.append((name, rc))
                print("[MultiTask] %s FAILED with exit code %d" % (name, rc), flush=True)
        running = still
        if running:
            time.sleep(interval)
    return failed


def main(argv=None):
    ap = argparse.ArgumentParser(description="Batch Compress")
    ap.add_argument("-stp", type=str, default=os.path.join(ROOT, "main.py"), help="the singletask script path")
    ap.add_argument("-p", type=str, default=os.path.join(ROOT, "opt", "MultiTask", "default.yaml"), help="yaml file path")
    ap.add_argument("-g", help="available gpu list", default="0", type=lambda s: [int(i) for i in s.split(",")])
    ap.add_argument("-t", type=float, default=2, help="the time interval between polls")
    ap.add_argument("-m", type=int, default=30, help="the max nums of task in running")
    ap.add_argument("-debug", action="store_true")
    ap.add_argument("-log", action="store_true")
    ap.add_argument("-onebyone", action="store_true")
    ap.add_argument("-dry", action="store_true", help="write the task YAMLs and print the commands, run nothing")
    args = ap.parse_args(argv)
    tasks, temp_dir = gen_task_list(args.p, args.stp)
    if args.dry:
        for name, cmd, path in tasks:
            print(name, " ".join(cmd))
        return 0
    try:
        failed = run(tasks, args.g, 1 if args.onebyone else args.m, args.t)
    finally:
        shutil.rmtree(temp_dir, ignore_errors=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
