long = int
