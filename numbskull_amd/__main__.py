from .numbskull import main

main()
